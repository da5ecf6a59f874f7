// cand_stats2.h -- V4 (mean_var_shift_polyA_check, reference adapted/detect/mvs.py:45-158) for ALL poly(A) candidates of a
// read in TWO sweeps over each array (round 4; cand_stats.h is the round-2 form: 4-5 sweeps, one LDS atomic per element and level,
// ~240 barriers per read -- kept behind ADP_CAND_STATS=old as the cross-check).
//
// What is asked (adapted/detect/combined.py:464, mvs.py:88-129): for every candidate c the five order statistics
// median(move_var(x)), median(move_mean(x)), median(x), percentile(x, 85), percentile(x, 15) of x = signal[adapter_end : cand_c].
// All slices start at the adapter end, the moving-window recurrences are causal: the candidates' arrays are PREFIXES of three
// arrays (the slice up to the largest candidate and its two series, k_mvs_series_wave).  The candidates' ends cut them into
// segments; every query is (array, prefix = segments 0 .. S, rank k).
//
//   sweep A  one running histogram per array, 2^HB bins of equal width over the key range of a SAMPLE of the array (the first and
//            the last bin take whatever lies outside: outliers do not widen the bins).  At the end of segment S the histogram is
//            that of the prefix: a block-wide prefix sum (each thread owns 2^HB / THREADS consecutive bins) places every rank that
//            ends there -- bin, elements below the bin, elements inside it -- and the sweep goes on.
//   slots    the distinct (array, bin) pairs among the queries; a bin shared by several prefixes keeps the LONGEST one's count.
//            The histogram words are dead by now and become the pool the bins' elements are copied to.
//   sweep B  an element whose bin has a slot (one byte of a look-up table per bin) and whose segment some query of that slot
//            covers is appended to the slot's list as (segment, key bits below the bin); minimum and maximum of a slot ride along.
//   finish   a wave per query: equal minimum and maximum (samples on an ADC grid) settle it; else the rank inside the bin by a
//            bit-wise search over the list (count of entries of segments <= S below a trial value: ballots).
//
// Exactness never rests on the sample: a rank that lands in one of the two open-ended bins, a pool that cannot hold a slot's list
// -- that candidate is left to k_validate's own statistics (ready = 0), as before.  Selection is exact; the float arithmetic on the
// selected values is cand_stats.h's (cs_pct_ranks / cs_pct_value = wave_percentile's).
#pragma once
#include "cand_stats.h"
#include "common.h"

extern __device__ int g_ablate; // (timing experiments, ADP_ABLATE: 2^20 no finish, 2^21 no sweep B, 2^23 no sweep A; tools/experiments/cs2_ablate.sh)
#define CS2_MAXC 16  // candidates per round (polya_cand_k beyond it: more rounds)
#define CS2_QPC 10   // queries per candidate: 6 on the slice (median pair, two percentile pairs), 2 + 2 on the series
#define CS2_SEGSH 27 // list entry = segment << 27 | key bits below the bin
#ifndef CS2_TPQ
#define CS2_TPQ 96   // lists up to this length are finished by one thread per query
#endif

template <int HB>
struct Cs2Sh {
    static constexpr int NB = 1 << HB;
    uint32_t hist[3 * NB]; // [array][bin]; after the slots are made: the pool of list entries
    uint8_t lut[3 * NB];   // [array][bin] -> slot (0xff: none)
    uint32_t wsum[3][16];  // per-wave totals of the block-wide prefix sum
    uint32_t klo[3], smin[3], smax[3];
    int shift[3], nan_first[3], wlen[3]; // wlen: window - 1 (what a series is shorter than the slice by)
    int nc, nseg, nslots, pool_used;
    int c_n[CS2_MAXC], c_seg[CS2_MAXC], c_bad[CS2_MAXC];
    int seglen[CS2_MAXC];
    uint32_t segmask[CS2_MAXC];
    int q_k[CS2_MAXC * CS2_QPC], q_bin[CS2_MAXC * CS2_QPC], q_before[CS2_MAXC * CS2_QPC], q_cnt[CS2_MAXC * CS2_QPC];
    float q_val[CS2_MAXC * CS2_QPC];
    int s_off[CS2_MAXC * CS2_QPC], s_total[CS2_MAXC * CS2_QPC], s_fill[CS2_MAXC * CS2_QPC], s_maxseg[CS2_MAXC * CS2_QPC], s_ok[CS2_MAXC * CS2_QPC];
    uint32_t s_min[CS2_MAXC * CS2_QPC], s_max[CS2_MAXC * CS2_QPC];
};

static __device__ __forceinline__ int cs2_arr(int t) { return t < 6 ? 0 : (t < 8 ? 1 : 2); }

// f(value, index) for every element of p[b0 .. b1): 16-byte loads (any alignment), CS2_U (template parameter) per thread in flight, all of them
// unconditional (the last chunk is read at b1 - 4 and its leading elements, which belong to the chunk before, are skipped)
template <int THREADS, int CS2_U, class F>
static __device__ __forceinline__ void cs2_sweep(const GLB float *p, int b0, int b1, int tid, F f)
{
    if (b1 - b0 < 4) {
        for (int i = b0 + tid; i < b1; i += THREADS) f(p[i], i);
        return;
    }
    const int last = b1 - 4;
    for (int base = b0 + tid * 4; base < b1; base += THREADS * 4 * CS2_U) {
        adp_f4u v[CS2_U];
#pragma unroll
        for (int u = 0; u < CS2_U; u++) {
            const int i = base + u * THREADS * 4;
            const int ic = i < last ? i : last;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const GLB adp_f4u *>(p + ic));
        }
#pragma unroll
        for (int u = 0; u < CS2_U; u++) {
            const int i = base + u * THREADS * 4;
            if (i < b1) {
                const int ic = i < last ? i : last;
                if (ic >= i) f(v[u].x, ic);
                if (ic + 1 >= i) f(v[u].y, ic + 1);
                if (ic + 2 >= i) f(v[u].z, ic + 2);
                f(v[u].w, ic + 3);
            }
        }
    }
}

// grid = n reads; block = THREADS; dynamic LDS = sizeof(Cs2Sh<HB>).  Arguments and results as k_cand_stats (cand_stats.h).
template <int THREADS, int HB, int CS2_U>
__global__ void __launch_bounds__(THREADS) k_cand_stats2(const float *__restrict__ sigs, const int32_t *__restrict__ full_len, int n_reads, int m,
                                                            const int64_t *__restrict__ bounds, int kmax, adp_cfg cfg,
                                                            const float *__restrict__ series, int cap, const int8_t *__restrict__ have,
                                                            CandStat *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cs2_raw[];
    typedef Cs2Sh<HB> Sh;
    constexpr int NB = Sh::NB, NW = THREADS / 64, BPT = NB / THREADS, POOL = 3 * NB, NQ = CS2_MAXC * CS2_QPC;
    static_assert(BPT >= 1 && BPT <= 8 && NW <= 16, "shape");
    LDS Sh *sh = (LDS Sh *)cs2_raw;
    const int r = blockIdx.x, tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    CandStat *o = out + (size_t)r * kmax;
    for (int c = tid; c < kmax; c += THREADS) o[c].ready = 0;
    if (!have[r]) return;
    const long long fl = full_len[r];
    const int S = (int)(fl < m ? fl : m);
    const int64_t *bd = bounds + (size_t)r * (1 + kmax);
    const long long a_e = bd[0];
    if (a_e == 0 || (long long)S < a_e + cfg.median_shift_window) return;
    const int a = (int)(a_e < S ? a_e : S);
    const int wvw = cfg.pA_var_window, wmw = cfg.pA_mean_window;
    const GLB float *arr[3];
    arr[0] = (const GLB float *)sigs + (size_t)r * m + a;
    arr[2] = (const GLB float *)series + (size_t)r * 2 * cap; // moving mean
    arr[1] = arr[2] + cap;                                    // moving variance
    int ncand = 0;
    while (ncand < kmax && bd[1 + ncand] != 0) ncand++;
    for (int c0 = 0; c0 < ncand; c0 += CS2_MAXC) {
        const int nc = ncand - c0 < CS2_MAXC ? ncand - c0 : CS2_MAXC;
        __syncthreads();
        // ---- the round's candidates, the segments their ends cut, their rank queries
        for (int i = tid; i < NQ; i += THREADS) { sh->q_k[i] = -1; sh->q_bin[i] = -1; sh->q_before[i] = 0; sh->q_cnt[i] = 0; sh->q_val[i] = 0.f; }
        if (tid < CS2_MAXC) { sh->segmask[tid] = 0; sh->c_bad[tid] = 0; sh->c_n[tid] = 0; sh->c_seg[tid] = -1; }
        if (tid < 3) { sh->smin[tid] = 0xffffffffu; sh->smax[tid] = 0u; sh->nan_first[tid] = 0x7fffffff; }
        if (tid == 0) { sh->wlen[0] = 0; sh->wlen[1] = wvw - 1; sh->wlen[2] = wmw - 1; sh->nslots = 0; sh->pool_used = 0; }
        __syncthreads();
        if (wv == 0) {
            int n = 0;
            if (ln < nc) {
                const long long p_e = bd[1 + c0 + ln];
                if (p_e >= a_e && p_e - a_e > 2 && p_e - a_e > wvw + 2 && p_e - a_e > wmw + 2) {
                    const int b = (int)(p_e < S ? p_e : S);
                    n = b - a;
                    if (wvw > n || wvw < 1 || wmw > n || wmw < 1 || n > cap) n = 0; // (an exception row, or no series: k_validate's business)
                }
            }
            bool first = n > 0;
            for (int c = 0; c < nc; c++) { const int n2 = __shfl(n, c); if (c < ln && n2 == n) first = false; }
            const unsigned long long fm = __ballot(first);
            int seg = 0;
            for (int c = 0; c < nc; c++) { const int n2 = __shfl(n, c); if (((fm >> c) & 1ull) && n2 < n) seg++; }
            if (ln == 0) { sh->nseg = __popcll(fm); sh->nc = nc; }
            if (ln < nc) {
                sh->c_n[ln] = n;
                if (n > 0) {
                    sh->c_seg[ln] = seg;
                    if (first) sh->seglen[seg] = n;
                    __hip_atomic_fetch_or(&sh->segmask[seg], 1u << ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    LDS int *qk = (LDS int *)sh->q_k + ln * CS2_QPC;
                    int lo, hi; double g;
                    qk[0] = n / 2; qk[1] = (n & 1) ? -1 : n / 2 - 1;
                    cs_pct_ranks(n, 85.0, lo, hi, g); qk[2] = hi; qk[3] = hi == lo ? -1 : hi - 1;
                    cs_pct_ranks(n, 15.0, lo, hi, g); qk[4] = hi; qk[5] = hi == lo ? -1 : hi - 1;
                    const int Lv = n - wvw + 1, Lm = n - wmw + 1;
                    qk[6] = Lv / 2; qk[7] = (Lv & 1) ? -1 : Lv / 2 - 1;
                    qk[8] = Lm / 2; qk[9] = (Lm & 1) ? -1 : Lm / 2 - 1;
                }
            }
        }
        __syncthreads();
        const int nseg = sh->nseg;
        if (nseg == 0) continue; // (uniform: every thread reads the same word)
        // ---- key range of each array from a sample: THREADS points spread over the whole array, THREADS over its first segment
        {
            const int n_all = sh->seglen[nseg - 1], n_first = sh->seglen[0];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int w1 = j == 0 ? 0 : (j == 1 ? wvw - 1 : wmw - 1);
                const int la = n_all - w1, lf = n_first - w1;
                const float v1 = arr[j][(int)(((long long)tid * la) / THREADS)], v2 = arr[j][(int)(((long long)tid * lf) / THREADS)];
                uint32_t mn = 0xffffffffu, mx = 0u;
                if (v1 == v1) { const uint32_t k = f2key(v1); mn = k; mx = k; }
                if (v2 == v2) { const uint32_t k = f2key(v2); mn = k < mn ? k : mn; mx = k > mx ? k : mx; }
                mn = wave_min(mn); mx = wave_max(mx);
                if (ln == 0) {
                    __hip_atomic_fetch_min(&sh->smin[j], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_max(&sh->smax[j], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        for (int i = tid; i < 3 * NB; i += THREADS) sh->hist[i] = 0;
        __syncthreads();
        if (tid < 3) {
            uint32_t mn = sh->smin[tid], mx = sh->smax[tid];
            if (mn > mx) { mn = 0; mx = 0; } // (a sample of NaNs only: any binning will do, the answers are NaN)
            const uint32_t span = mx - mn;
            int s = 0;
            while ((span >> s) + 1u > (uint32_t)(NB - 2)) s++;
            // bin b (1 <= b <= NB - 2) = keys [klo + b 2^s, klo + (b + 1) 2^s); bin 0 and bin NB - 1 are open-ended
            sh->shift[tid] = s;
            sh->klo[tid] = mn >= (1u << s) ? mn - (1u << s) : 0u;
        }
        __syncthreads();
        uint32_t klo[3]; int shf[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { klo[j] = sh->klo[j]; shf[j] = sh->shift[j]; }
        // ---- sweep A: running histograms, the ranks of every prefix placed at its end
        int nanmin[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        {
            int prev = 0;
            for (int s = 0; s < nseg; s++) {
                const int hi = sh->seglen[s];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int w1 = j == 0 ? 0 : (j == 1 ? wvw - 1 : wmw - 1);
                    const int b0 = prev - w1 > 0 ? prev - w1 : 0, b1 = hi - w1;
                    LDS uint32_t *h = (LDS uint32_t *)sh->hist + j * NB;
                    const uint32_t kl = klo[j]; const int sf = shf[j];
                    int nm = nanmin[j];
                    if (!(g_ablate & (1 << 23)))
                    cs2_sweep<THREADS, CS2_U>(arr[j], b0, b1, tid, [&](float v, int i) {
                        if (v != v) { nm = i < nm ? i : nm; return; }
                        const uint32_t key = f2key(v);
                        uint32_t b = ((key > kl ? key : kl) - kl) >> sf;
                        b = b < (uint32_t)(NB - 1) ? b : (uint32_t)(NB - 1);
                        __hip_atomic_fetch_add(&h[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    });
                    nanmin[j] = nm;
                }
                prev = hi;
                __syncthreads();
                // block-wide prefix sums of the three histograms: thread t owns bins [t BPT, (t + 1) BPT)
                uint32_t bins[3][BPT], sum[3], excl[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const LDS uint32_t *h = (const LDS uint32_t *)sh->hist + j * NB + tid * BPT;
                    uint32_t t = 0;
#pragma unroll
                    for (int e = 0; e < BPT; e++) { bins[j][e] = h[e]; t += bins[j][e]; }
                    sum[j] = t;
                    const uint32_t incl = (uint32_t)wave_scan_incl((int)t);
                    excl[j] = incl - t;
                    if (ln == 63) sh->wsum[j][wv] = incl;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    uint32_t off = 0;
                    for (int w = 0; w < NW; w++) { const uint32_t t = sh->wsum[j][w]; off += w < wv ? t : 0u; }
                    excl[j] += off;
                }
                uint32_t cm = sh->segmask[s];
                while (cm) {
                    const int c = __ffs((int)cm) - 1;
                    cm &= cm - 1;
#pragma unroll
                    for (int t = 0; t < CS2_QPC; t++) {
                        const int k = sh->q_k[c * CS2_QPC + t];
                        const int j = cs2_arr(t);
                        if (k >= 0 && (uint32_t)k >= excl[j] && (uint32_t)k < excl[j] + sum[j]) {
                            uint32_t acc = excl[j];
                            int eb = 0; uint32_t ecnt = 0, ebefore = 0; bool found = false;
#pragma unroll
                            for (int e = 0; e < BPT; e++) {
                                const uint32_t cnt = bins[j][e];
                                if (!found && (uint32_t)k < acc + cnt) { found = true; eb = e; ecnt = cnt; ebefore = acc; }
                                acc += cnt;
                            }
                            sh->q_bin[c * CS2_QPC + t] = tid * BPT + eb;
                            sh->q_before[c * CS2_QPC + t] = (int)ebefore;
                            sh->q_cnt[c * CS2_QPC + t] = (int)ecnt;
                        }
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int nm = wave_min(nanmin[j]);
            if (ln == 0 && nm != 0x7fffffff) __hip_atomic_fetch_min(&sh->nan_first[j], nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // ---- slots: one per distinct (array, bin) among the queries; the histogram word of such a bin first becomes a mail box
        // (largest segment of any query on the bin, with that prefix's count), then the pool takes the histograms' place
        for (int i = tid; i < 3 * NB / 4; i += THREADS) ((LDS uint32_t *)sh->lut)[i] = 0xffffffffu;
        int my_q = -1, my_j = 0, my_bin = 0;
        uint32_t my_code = 0;
        if (tid < NQ) {
            const int c = tid / CS2_QPC, t = tid % CS2_QPC;
            if (c < nc && sh->q_k[tid] >= 0) {
                const int b = sh->q_bin[tid];
                my_j = cs2_arr(t);
                if (b >= 1 && b <= NB - 2 && sh->q_cnt[tid] < (1 << 20)) {
                    my_q = tid; my_bin = b;
                    my_code = ((uint32_t)(sh->c_seg[c] + 1) << 20) | (uint32_t)sh->q_cnt[tid];
                    sh->hist[my_j * NB + b] = 0u;
                } else sh->c_bad[c] = 1; // an open-ended bin (or no bin: more NaNs than the rank allows): not settled here
            }
        }
        __syncthreads();
        if (my_q >= 0) __hip_atomic_fetch_max(&sh->hist[my_j * NB + my_bin], my_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        if (my_q >= 0) {
            LDS uint32_t *w = (LDS uint32_t *)&sh->hist[my_j * NB + my_bin];
            if ((*w & 0x7fffffffu) == my_code) {
                const uint32_t old = __hip_atomic_fetch_or(w, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!(old >> 31)) { // the first of the winners makes the slot
                    const int cnt = (int)(my_code & 0xfffffu);
                    const int pos = __hip_atomic_fetch_add(&sh->pool_used, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int sl = __hip_atomic_fetch_add(&sh->nslots, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    sh->s_off[sl] = pos; sh->s_total[sl] = cnt; sh->s_fill[sl] = 0; sh->s_maxseg[sl] = (int)(my_code >> 20) - 1;
                    sh->s_ok[sl] = (pos + cnt <= POOL) ? 1 : 0;
                    sh->s_min[sl] = 0xffffffffu; sh->s_max[sl] = 0u;
                    sh->lut[my_j * NB + my_bin] = (uint8_t)sl;
                }
            }
        }
        __syncthreads();
        // ---- sweep B: the elements of the slots' bins, up to the longest prefix that asks for them
        {
            LDS uint32_t *pool = (LDS uint32_t *)sh->hist;
            int prev = 0;
            for (int s = 0; s < nseg; s++) {
                const int hi = sh->seglen[s];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int w1 = j == 0 ? 0 : (j == 1 ? wvw - 1 : wmw - 1);
                    const int b0 = prev - w1 > 0 ? prev - w1 : 0, b1 = hi - w1;
                    const LDS uint8_t *lut = (const LDS uint8_t *)sh->lut + j * NB;
                    const uint32_t kl = klo[j]; const int sf = shf[j];
                    const uint32_t lowmask = sf ? ((1u << sf) - 1u) : 0u;
                    if (!(g_ablate & (1 << 21)))
                    cs2_sweep<THREADS, CS2_U>(arr[j], b0, b1, tid, [&](float v, int i) {
                        (void)i;
                        if (v != v) return;
                        const uint32_t key = f2key(v);
                        const uint32_t d = (key > kl ? key : kl) - kl;
                        uint32_t b = d >> sf;
                        b = b < (uint32_t)(NB - 1) ? b : (uint32_t)(NB - 1);
                        const int sl = lut[b];
                        if (sl == 0xff) return;
                        if (s > sh->s_maxseg[sl]) return;
                        const uint32_t low = d & lowmask;
                        __hip_atomic_fetch_min(&sh->s_min[sl], low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_max(&sh->s_max[sl], low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (sh->s_ok[sl]) {
                            const int p = __hip_atomic_fetch_add(&sh->s_fill[sl], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (p < sh->s_total[sl]) pool[sh->s_off[sl] + p] = ((uint32_t)s << CS2_SEGSH) | low;
                        }
                    });
                }
                prev = hi;
            }
        }
        __syncthreads();
        // ---- finish, short lists: a THREAD per query (round 4: at the default window a bin holds a few dozen elements, and the wave-per-query
        // loop below -- a ballot per bit and 64 entries, 25 queries in turn per wave -- was a third of the kernel); the query is done when its
        // rank comes back as -1
        if (THREADS <= 256 && !(g_ablate & (1 << 20))) { // (the small shape: at the 200 k window the lists are long and this phase only adds to the wave loop: 15.0 -> 15.7 ms)
            const LDS uint32_t *pool = (const LDS uint32_t *)sh->hist;
            for (int qi = tid; qi < nc * CS2_QPC; qi += THREADS) {
                const int k = sh->q_k[qi];
                if (k < 0) continue;
                const int c = qi / CS2_QPC, t = qi % CS2_QPC, j = cs2_arr(t);
                const int b = sh->q_bin[qi];
                if (b < 1 || b > NB - 2 || sh->q_cnt[qi] >= (1 << 20)) continue; // (the wave loop's business: c_bad is set there)
                const int sl = sh->lut[j * NB + b];
                if (sl == 0xff) continue;
                const uint32_t smn = sh->s_min[sl], smx = sh->s_max[sl];
                uint32_t low;
                if (smn == smx) low = smn;
                else {
                    const int total = sh->s_total[sl];
                    if (!sh->s_ok[sl] || total > CS2_TPQ) continue;
                    const int Sq = sh->c_seg[c], krem = k - sh->q_before[qi];
                    const LDS uint32_t *list = pool + sh->s_off[sl];
                    uint32_t pre = 0;
                    for (int bit = shf[j] - 1; bit >= 0; bit--) {
                        const uint32_t trial = pre | (1u << bit);
                        int cnt = 0;
                        for (int i = 0; i < total; i++) {
                            const uint32_t e = list[i];
                            cnt += ((int)(e >> CS2_SEGSH) <= Sq && (e & ((1u << CS2_SEGSH) - 1u)) < trial) ? 1 : 0;
                        }
                        if (cnt <= krem) pre = trial;
                    }
                    low = pre;
                }
                const uint32_t key = klo[j] + ((uint32_t)b << shf[j]) + low;
                const int len = sh->c_n[c] - sh->wlen[j];
                sh->q_val[qi] = (len > sh->nan_first[j]) ? __builtin_nanf("") : key2f(key);
                sh->q_k[qi] = -1;
            }
        }
        __syncthreads();
        // ---- finish: a wave per query (the long lists)
        {
            const LDS uint32_t *pool = (const LDS uint32_t *)sh->hist;
            for (int qi = wv; qi < nc * CS2_QPC; qi += NW) {
                const int k = sh->q_k[qi];
                if (k < 0 || (g_ablate & (1 << 20))) continue;
                const int c = qi / CS2_QPC, t = qi % CS2_QPC, j = cs2_arr(t);
                const int b = sh->q_bin[qi];
                if (b < 1 || b > NB - 2 || sh->q_cnt[qi] >= (1 << 20)) continue; // (c_bad is set)
                const int sl = sh->lut[j * NB + b];
                if (sl == 0xff) { if (ln == 0) sh->c_bad[c] = 1; continue; }
                uint32_t low;
                const uint32_t smn = sh->s_min[sl], smx = sh->s_max[sl];
                if (smn == smx) low = smn;
                else if (!sh->s_ok[sl]) { if (ln == 0) sh->c_bad[c] = 1; continue; }
                else {
                    const int total = sh->s_total[sl], Sq = sh->c_seg[c];
                    const int krem = k - sh->q_before[qi];
                    const LDS uint32_t *list = pool + sh->s_off[sl];
                    uint32_t pre = 0;
                    for (int bit = shf[j] - 1; bit >= 0; bit--) {
                        const uint32_t trial = pre | (1u << bit);
                        int cnt = 0;
                        for (int i0 = 0; i0 < total; i0 += 64) {
                            const int i = i0 + ln;
                            const uint32_t e = i < total ? list[i] : 0xffffffffu;
                            const bool below = (int)(e >> CS2_SEGSH) <= Sq && (e & ((1u << CS2_SEGSH) - 1u)) < trial;
                            cnt += __popcll(__ballot(below));
                        }
                        if (cnt <= krem) pre = trial;
                    }
                    low = pre;
                }
                if (ln == 0) {
                    const uint32_t key = klo[j] + ((uint32_t)b << shf[j]) + low;
                    const int len = sh->c_n[c] - sh->wlen[j];
                    sh->q_val[qi] = (len > sh->nan_first[j]) ? __builtin_nanf("") : key2f(key);
                }
            }
        }
        __syncthreads();
        if (tid < nc && sh->c_n[tid] > 0 && !sh->c_bad[tid]) {
            const int c = tid, n = sh->c_n[c];
            const LDS float *qv = (const LDS float *)sh->q_val + c * CS2_QPC;
            CandStat st;
            const int Lv = n - wvw + 1, Lm = n - wmw + 1;
            st.fvar = (Lv & 1) ? qv[6] : (qv[7] + qv[6]) / 2.0f;
            st.fmean = (Lm & 1) ? qv[8] : (qv[9] + qv[8]) / 2.0f;
            st.fmed = (n & 1) ? qv[0] : (qv[1] + qv[0]) / 2.0f;
            int lo, hi; double g;
            cs_pct_ranks(n, 85.0, lo, hi, g); st.q85 = cs_pct_value(qv[2], qv[3], lo, hi, g);
            cs_pct_ranks(n, 15.0, lo, hi, g); st.q15 = cs_pct_value(qv[4], qv[5], lo, hi, g);
            st.ready = 1;
            o[c0 + c] = st;
        }
    }
}
