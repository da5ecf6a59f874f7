// cand_stats.h -- V4 (mean_var_shift_polyA_check, reference adapted/detect/mvs.py:45-158) for ALL poly(A) candidates of a
// read in shared passes.
//
// The CNN path hands validate_boundaries up to k = 10 candidates per read (adapted/detect/combined.py:464); the reference
// checks them one after the other, each check taking five order statistics of slices that all START at the adapter end:
// median(move_var(x)), median(move_mean(x)), median(x), percentile(x, 85), percentile(x, 15) with x = signal[adapter_end :
// candidate].  At the 200 k window a slice is up to ~190 k samples and a failing read runs all ten: one wave doing 50
// four-pass radix selects over them was 70 % of the CNN step.  But the slices are PREFIXES of one array (and so are the
// moving-window series: the recurrences are causal), so one sweep serves every candidate:
//
//   * the candidates' ends cut the array into segments; a running histogram of the keys is complete for a candidate when
//     the sweep reaches its end -- its ranks are located right then, and the sweep goes on;
//   * level 0 resolves 12 bits of the key span, the next levels 8 bits each, for all (candidate, rank) queries at once:
//     an element is routed through small LDS tables to the histogram of the queries whose resolved bits it shares;
//   * x_(k-1) (even-count medians, percentile interpolation) is simply another query.
//
// A workgroup per read; four sweeps over each of the three arrays (min / max, then 12 + 8 + 8 bits; a fifth for spans beyond
// 28 bits; the small shape: 8 + 6 + 6 + 6 + 6) whatever the number of candidates.  Selection is exact; the float32 / float64 arithmetic on the
// selected values is that of wave_median / wave_percentile (wave_stats.h).
#pragma once
#include "common.h"

// Two shapes of the same kernel (template parameters THREADS, L0, LN = bits of the key span resolved by the first / by a later
// level): 1024 threads with 12 + 8 bits for the long slices of wide windows, where the sweeps over the arrays are the cost
// and levels must be few; 256 threads with 8 + 6 bits for the short slices of the default window, where the work per
// level that grows with the bins (clearing, slot maps, rank scans) and the barriers of a big workgroup are the cost
// (per 32 000 reads at the 16 k window: 1024 / 12 / 8 54 ms, 256 / 12 / 8 62, 256 / 8 / 6 9.1 with the fifth level wide
// key spans need; k_validate's own wave-per-read statistics: 12.3).
#define CS_MAXQ 64
#ifndef CS_ADAPT_SHAPES
// which shapes widen their later levels to the histogram's capacity (1 = the 1024-thread one, 2 = the 256-thread one), and the
// first level of the big shape.  Measured (8000 reads of the 200 k window / 32 000 of the 16 k window, one box): 12 bits, fixed
// widths 19.9 ms; 12 bits, adaptive 21.0 (one sweep of four less, but a wave's rank scan over 2^11 bins per slot and segment
// costs more than the sweep); 14 bits 45 ms (histogram atomics over 64 KB, 256 bins per lane in every rank scan); the small
// shape 9.02 -> 8.44 ms with adaptive widths.  The sweeps are not what bounds the big shape.
#define CS_ADAPT_SHAPES 2
#endif
#ifndef CS_BIG_L0
#define CS_BIG_L0 12
#endif
#define CS_GROUP 10 // candidates per round (6 queries each on the slice itself)
#define CS_UNROLL 8 // elements a thread requests before it looks at the first (memory-level parallelism of the sweeps)

struct CandStat { float fvar, fmean, fmed; int32_t ready; double q85, q15; };

struct CsQuery { int len, k, slot, krem, bin, before; uint32_t prefix; float val; };

template <int L0, int LN>
struct CsSharedT {
    static_assert((1 << L0) <= (CS_MAXQ << LN), "the first level uses the histogram words of all slots");
    uint32_t hist[CS_MAXQ << LN];
    static constexpr int NLEV = 1 + (32 - L0 + LN - 1) / LN; // levels that resolve any 32-bit key span (4 for 12 + 8, 5 for 8 + 6)
    uint8_t map0[1 << L0];
    uint8_t mapn[NLEV - 2][CS_MAXQ << LN]; // slot routing behind the levels 1 .. NLEV - 2
    CsQuery q[CS_MAXQ];
    int seglen[CS_MAXQ];
    int nseg, nq, nslots;
    uint32_t mn, mx;
    int nan_first;
    // per candidate of the round
    int c_n[CS_GROUP], c_qx[CS_GROUP][6], c_qv[CS_GROUP][2], c_qm[CS_GROUP][2];
    float c_x[CS_GROUP][6], c_v[CS_GROUP][2], c_m[CS_GROUP][2];
};

// x_(k) of x[0 .. len) for every query sh->q[0 .. nq) (0 <= k < len <= n_max); NaN when the prefix holds one.
// All threads of the block call it; results in sh->q[i].val.
template <int THREADS, int L0, int LN>
static __device__ void cs_multi_select(const float *__restrict__ x_, LDS CsSharedT<L0, LN> *sh)
{
    constexpr int CS_THREADS = THREADS, CS_L0_BITS = L0, CS_LN_BITS = LN;
    constexpr bool CS_ADAPT = CS_ADAPT_SHAPES & (THREADS >= 1024 ? 1 : 2);
    const GLB float *x = (const GLB float *)x_;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    const int nq = sh->nq;
    if (nq == 0) return;
    if (tid == 0) { // the distinct prefix lengths, ascending
        int ns = 0;
        for (int i = 0; i < nq; i++) {
            const int L = sh->q[i].len;
            int j = ns;
            while (j > 0 && sh->seglen[j - 1] > L) j--;
            if (j > 0 && sh->seglen[j - 1] == L) continue;
            for (int t = ns; t > j; t--) sh->seglen[t] = sh->seglen[t - 1];
            sh->seglen[j] = L; ns++;
        }
        sh->nseg = ns;
        sh->mn = 0xffffffffu; sh->mx = 0u; sh->nan_first = 0x7fffffff;
    }
    __syncthreads();
    const int nseg = sh->nseg;
    const int n_all = sh->seglen[nseg - 1];
    {   // smallest / largest key, first NaN
        uint32_t mn = 0xffffffffu, mx = 0u; int nf = 0x7fffffff;
        for (int base = tid; base < n_all; base += CS_UNROLL * CS_THREADS) { // (the loads of a round go out together)
            float v[CS_UNROLL];
#pragma unroll
            for (int u = 0; u < CS_UNROLL; u++) { const int i = base + u * CS_THREADS; v[u] = i < n_all ? x[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < CS_UNROLL; u++) {
                const int i = base + u * CS_THREADS;
                if (i >= n_all) break;
                if (v[u] != v[u]) { if (i < nf) nf = i; }
                else { const uint32_t key = f2key(v[u]); mn = key < mn ? key : mn; mx = key > mx ? key : mx; }
            }
        }
        mn = wave_min(mn); mx = wave_max(mx); nf = wave_min(nf);
        if (ln == 0) {
            __hip_atomic_fetch_min(&sh->mn, mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(&sh->mx, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_min(&sh->nan_first, nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const uint32_t mn = sh->mn;
    const uint32_t span = sh->mx >= mn ? sh->mx - mn : 0u;
    const int nan_first = sh->nan_first;
    for (int i = tid; i < nq; i += CS_THREADS) { sh->q[i].slot = 0; sh->q[i].krem = sh->q[i].k; sh->q[i].prefix = 0; }
    int rb = span ? 32 - __clz(span) : 0;
    constexpr int NLEV = CsSharedT<L0, LN>::NLEV;
    int sft[NLEV], wid[NLEV]; // shifts / widths of the levels already resolved (the routing below)
#pragma unroll
    for (int l = 0; l < NLEV; l++) sft[l] = wid[l] = 0;
    int nslots = 1;
#pragma unroll
    for (int level = 0; level < NLEV; level++) {
        if (rb <= 0) break;
        // a later level is as wide as the histogram words allow for the slots the queries still occupy (their ranks mostly share
        // a few bins of the level before): 14 + 11 bits settle a 25-bit key span in two sweeps where 12 + 8 + 5 took three
        int wcap = CS_LN_BITS;
        if (CS_ADAPT && level > 0) { const int room = (CS_MAXQ << CS_LN_BITS) / nslots; while ((2 << wcap) <= room && wcap < 16) wcap++; }
        const int w = level == 0 ? (rb < CS_L0_BITS ? rb : CS_L0_BITS) : (rb < wcap ? rb : wcap);
        const int shift = rb - w;
        sft[level] = shift; wid[level] = w;
        for (int i = tid; i < (nslots << w); i += CS_THREADS) sh->hist[i] = 0;
        __syncthreads();
        int lo = 0;
        for (int s = 0; s < nseg; s++) {
            const int hi = sh->seglen[s];
            for (int base = lo + tid; base < hi; base += CS_UNROLL * CS_THREADS) {
                float vv[CS_UNROLL];
#pragma unroll
                for (int u = 0; u < CS_UNROLL; u++) { const int i = base + u * CS_THREADS; vv[u] = i < hi ? x[i] : __builtin_nanf(""); }
#pragma unroll
                for (int u = 0; u < CS_UNROLL; u++) {
                    const float v = vv[u];
                    if (v != v) continue;
                    const uint32_t d = f2key(v) - mn;
                    uint32_t slot = 0;
                    if (level >= 1) { slot = sh->map0[d >> sft[0]]; if (slot == 0xffu) continue; }
                    bool dead = false;
#pragma unroll
                    for (int l = 1; l < NLEV - 1; l++) {
                        if (l < level && !dead) {
                            slot = sh->mapn[l - 1][(slot << wid[l]) | ((d >> sft[l]) & ((1u << wid[l]) - 1u))];
                            dead = slot == 0xffu;
                        }
                    }
                    if (dead) continue;
                    __hip_atomic_fetch_add(&sh->hist[(slot << w) | ((d >> shift) & ((1u << w) - 1u))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            lo = hi;
            __syncthreads();
            // the queries that end here: the bin of their rank (a wave per query; a lane owns 2^w / 64 consecutive bins)
            for (int qi = wv; qi < nq; qi += CS_THREADS / 64) {
                if (sh->q[qi].len != hi) continue;
                const LDS uint32_t *h = sh->hist + ((size_t)sh->q[qi].slot << w);
                const int nb = 1 << w, per = (nb + 63) / 64;
                const int b0 = ln * per, b1 = (b0 + per < nb) ? b0 + per : nb;
                int sum = 0;
                for (int b = b0; b < b1; b++) sum += (int)h[b];
                const int incl = wave_scan_incl(sum), excl = incl - sum;
                const int kr = sh->q[qi].krem;
                const bool mine = kr >= excl && kr < incl;
                if (mine) {
                    int acc = excl, b = b0;
                    for (; b < b1; b++) { const int c = (int)h[b]; if (kr < acc + c) break; acc += c; }
                    sh->q[qi].bin = b; sh->q[qi].before = acc;
                }
            }
            __syncthreads();
        }
        // next level: one slot per distinct (slot, bin) among the queries
        const int last = (shift == 0);
        if (!last) {
            LDS uint8_t *map = level == 0 ? (LDS uint8_t *)sh->map0 : (LDS uint8_t *)sh->mapn[level - 1];
            for (int i = tid; i < (nslots << w); i += CS_THREADS) map[i] = 0xff;
            __syncthreads();
            if (tid == 0) {
                int D = 0;
                for (int i = 0; i < nq; i++) {
                    const int key = (sh->q[i].slot << w) | sh->q[i].bin;
                    if (map[key] == 0xff) map[key] = (uint8_t)D++;
                }
                sh->nslots = D;
            }
            __syncthreads();
            nslots = sh->nslots;
            for (int i = tid; i < nq; i += CS_THREADS) {
                const int key = (sh->q[i].slot << w) | sh->q[i].bin;
                sh->q[i].prefix = (sh->q[i].prefix << w) | (uint32_t)sh->q[i].bin;
                sh->q[i].krem -= sh->q[i].before;
                sh->q[i].slot = map[key];
            }
        } else {
            for (int i = tid; i < nq; i += CS_THREADS) sh->q[i].prefix = (sh->q[i].prefix << w) | (uint32_t)sh->q[i].bin;
        }
        __syncthreads();
        rb = shift;
    }
    for (int i = tid; i < nq; i += CS_THREADS)
        sh->q[i].val = (sh->q[i].len > nan_first) ? __builtin_nanf("") : key2f(mn + sh->q[i].prefix);
    __syncthreads();
}

// np.percentile (linear) from x_(hi) = vk and x_(hi - 1) = vkm1: the arithmetic of wave_percentile
static __device__ __forceinline__ void cs_pct_ranks(int n, double q100, int &lo, int &hi, double &g)
{
    const double vi = (double)(n - 1) * (q100 / 100.0);
    lo = (int)floor(vi);
    if (lo < 0) lo = 0;
    if (lo > n - 1) lo = n - 1;
    hi = lo + 1 < n - 1 ? lo + 1 : n - 1;
    g = vi - (double)lo;
}
static __device__ __forceinline__ double cs_pct_value(float vk, float vkm1, int lo, int hi, double g)
{
    const float a = (hi == lo) ? vk : vkm1, b = vk;
    const float diff = b - a;
    double r = (double)a + (double)diff * g;
    if (g >= 0.5) r = (double)b - (double)diff * (1.0 - g);
    return r;
}

// grid = n reads; block = THREADS; dynamic LDS = sizeof(CsSharedT<L0, LN>).  series: [n, 2, cap] moving mean, moving variance of
// signal[adapter_end : largest candidate) (k_mvs_series); out: [n, kmax], ready = 1 where the five statistics are filled in
// (k_validate computes the others itself: slices shorter than a window, reads whose series were not prepared).
template <int THREADS, int L0, int LN>
__global__ void __launch_bounds__(THREADS) k_cand_stats(const float *__restrict__ sigs, const int32_t *__restrict__ full_len, int n_reads, int m,
                                                           const int64_t *__restrict__ bounds, int kmax, adp_cfg cfg,
                                                           const float *__restrict__ series, int cap, const int8_t *__restrict__ have,
                                                           CandStat *__restrict__ out)
{
    extern __shared__ unsigned char cs_raw[];
    typedef CsSharedT<L0, LN> CsShared;
    LDS CsShared *sh = (LDS CsShared *)cs_raw;
    const int r = blockIdx.x, tid = threadIdx.x;
    CandStat *o = out + (size_t)r * kmax;
    if (tid < kmax) o[tid].ready = 0;
    if (!have[r]) return;
    const long long fl = full_len[r];
    const int S = (int)(fl < m ? fl : m);
    const int64_t *bd = bounds + (size_t)r * (1 + kmax);
    const long long a_e = bd[0];
    if (a_e == 0 || (long long)S < a_e + cfg.median_shift_window) return;
    const int a = (int)(a_e < S ? a_e : S);
    const int wvw = cfg.pA_var_window, wmw = cfg.pA_mean_window;
    const float *x = sigs + (size_t)r * m + a;
    const float *smean = series + (size_t)r * 2 * cap, *svar = smean + cap;
    int ncand = 0;
    while (ncand < kmax && bd[1 + ncand] != 0) ncand++;
    for (int c0 = 0; c0 < ncand; c0 += CS_GROUP) {
        const int nc = ncand - c0 < CS_GROUP ? ncand - c0 : CS_GROUP;
        __syncthreads();
        if (tid == 0) { // the round's candidates and their rank queries on the slice
            int nq = 0;
            for (int c = 0; c < nc; c++) {
                const long long p_e = bd[1 + c0 + c];
                int n = 0;
                if (p_e >= a_e && p_e - a_e > 2 && p_e - a_e > wvw + 2 && p_e - a_e > wmw + 2) {
                    const int b = (int)(p_e < S ? p_e : S);
                    n = b - a;
                    if (wvw > n || wvw < 1 || wmw > n || wmw < 1 || n > cap) n = 0; // (an exception row, or no series: k_validate's business)
                }
                sh->c_n[c] = n;
                if (n <= 0) continue;
                int lo, hi; double g;
                int ks[6]; ks[0] = n / 2; ks[1] = (n & 1) ? -1 : n / 2 - 1;
                cs_pct_ranks(n, 85.0, lo, hi, g); ks[2] = hi; ks[3] = hi == lo ? -1 : hi - 1;
                cs_pct_ranks(n, 15.0, lo, hi, g); ks[4] = hi; ks[5] = hi == lo ? -1 : hi - 1;
                for (int t = 0; t < 6; t++) {
                    sh->c_qx[c][t] = -1;
                    if (ks[t] >= 0) { sh->q[nq].len = n; sh->q[nq].k = ks[t]; sh->c_qx[c][t] = nq++; }
                }
            }
            sh->nq = nq;
        }
        __syncthreads();
        cs_multi_select<THREADS, L0, LN>(x, sh);
        if (tid < nc * 6) { const int c = tid / 6, t = tid % 6; const int qi = sh->c_qx[c][t]; sh->c_x[c][t] = (sh->c_n[c] > 0 && qi >= 0) ? sh->q[qi].val : 0.f; }
        __syncthreads();
        for (int which = 0; which < 2; which++) { // the moving variance, then the moving mean: median of the prefix n - w + 1
            const int w = which == 0 ? wvw : wmw;
            if (tid == 0) {
                int nq = 0;
                for (int c = 0; c < nc; c++) {
                    LDS int *cq = which == 0 ? (LDS int *)sh->c_qv[c] : (LDS int *)sh->c_qm[c];
                    cq[0] = cq[1] = -1;
                    const int n = sh->c_n[c];
                    if (n <= 0) continue;
                    const int L = n - w + 1;
                    sh->q[nq].len = L; sh->q[nq].k = L / 2; cq[0] = nq++;
                    if (!(L & 1)) { sh->q[nq].len = L; sh->q[nq].k = L / 2 - 1; cq[1] = nq++; }
                }
                sh->nq = nq;
            }
            __syncthreads();
            cs_multi_select<THREADS, L0, LN>(which == 0 ? svar : smean, sh);
            if (tid < nc * 2) {
                const int c = tid / 2, t = tid % 2;
                const int qi = which == 0 ? sh->c_qv[c][t] : sh->c_qm[c][t];
                const float v = (sh->c_n[c] > 0 && qi >= 0) ? sh->q[qi].val : 0.f;
                if (which == 0) sh->c_v[c][t] = v; else sh->c_m[c][t] = v;
            }
            __syncthreads();
        }
        if (tid < nc && sh->c_n[tid] > 0) {
            const int c = tid, n = sh->c_n[c];
            CandStat st;
            const int Lv = n - wvw + 1, Lm = n - wmw + 1;
            st.fvar = (Lv & 1) ? sh->c_v[c][0] : (sh->c_v[c][1] + sh->c_v[c][0]) / 2.0f;
            st.fmean = (Lm & 1) ? sh->c_m[c][0] : (sh->c_m[c][1] + sh->c_m[c][0]) / 2.0f;
            st.fmed = (n & 1) ? sh->c_x[c][0] : (sh->c_x[c][1] + sh->c_x[c][0]) / 2.0f;
            int lo, hi; double g;
            cs_pct_ranks(n, 85.0, lo, hi, g); st.q85 = cs_pct_value(sh->c_x[c][2], sh->c_x[c][3], lo, hi, g);
            cs_pct_ranks(n, 15.0, lo, hi, g); st.q15 = cs_pct_value(sh->c_x[c][4], sh->c_x[c][5], lo, hi, g);
            st.ready = 1;
            o[c0 + c] = st;
        }
    }
}
