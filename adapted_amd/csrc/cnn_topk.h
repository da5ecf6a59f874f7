// cnn_topk.h -- C3: cnn_predict on the device (reference adapted/detect/cnn.py:101-160).
//
//   adapter_pos = argmax(scores[:, 0, :na]);  mask scores[:, 1, :adapter_pos] to -5;  polya_pos = argmax(scores[:, 1, :]);
//   k > 1: mask scores[:, 1, polya_pos + 1:] to -5, scipy.signal.find_peaks(distance=5) on the FLATTENED [n * Lo] array,
//   group the peaks by read (flat index // Lo), order each group by descending height (ties: ascending index), keep k;
//   the group of the i-th read THAT HAS PEAKS is written into row i (cnn.py:150-158: a read without peaks shifts the
//   later groups up); finally index * ds + min_obs_adapter, and a result equal to min_obs_adapter becomes 0 (:173-179).
//
// Kernels: k_cnn_argmax (a wave per read: both arg-maxes, numpy's rules: first occurrence, a NaN wins),
// k_cnn_rowlink (one block: nearest read at or before r that has unmasked samples), k_cnn_topk (a wave per CHAIN of
// reads, normally one read), k_cnn_bounds (one block: compaction + scaling into the validator's candidate table).
//
// find_peaks on the flattened, masked array, restated (scipy/signal/_peak_finding_utils.pyx: _local_maxima_1d,
// _select_by_peak_distance):
//   * a peak is a sample, or the midpoint (left + right) // 2 of a run of equal samples, whose neighbours on both sides
//     exist and are lower.  Between the unmasked stretches [a_r, p_r] of consecutive reads lies a run of masked samples
//     (-5): the reads of a batch only meet across it in corner cases -- a stretch that ends within 5 samples of its row's
//     end next to one that starts within 5 samples of the next row's start, or a read all of whose scores lie at or below
//     the mask level with nothing masked in front (its masked run may then be a peak itself, or merge with samples that
//     equal -5 exactly).  Rows whose common boundary shows none of this are DECOUPLED (ct_decoupled); maximal runs of
//     coupled rows form a chain, handled as one piece of the flat array by the wave of its first row.  A chain's ends
//     are masked runs that are not peaks, at least 5 samples long on either side, so a chain is self-contained.
//   * minimum distance 5: peaks are visited by descending height and a visited, still kept peak removes every peak
//     closer than 5 samples.  Local maxima are at least 2 apart, so at most two lie that close on each side: the rule is
//     the fixed point of "kept iff no kept higher neighbour" on the ordinals of the maxima.  Equal heights: scipy takes
//     the order from an unstable np.argsort (no defined order beyond 16 elements); here the LATER index counts as higher,
//     which is what a stable sort gives and what the insertion sort numpy uses on short arrays gives.
#pragma once
#include "common.h"

#define CT_EXCL (-5.0f)

// ---------------------------------------------------------------- arg-maxes
// np.argmax: first occurrence of the maximum; a NaN counts as the maximum (first NaN wins)
struct CtBest { float v; int i; };
static __device__ __forceinline__ bool ct_better(float v, int i, float bv, int bi)
{
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;
    if (vn) return i < bi;
    return v > bv || (v == bv && i < bi);
}
static __device__ __forceinline__ CtBest ct_wave_argmax(const float *__restrict__ x, int lo, int hi) // [lo, hi), hi > lo
{
    const int ln = lane_id();
    float bv = -__builtin_inff(); int bi = 0x7fffffff;
    for (int j = lo + ln; j < hi; j += 64) { const float v = x[j]; if (ct_better(v, j, bv, bi)) { bv = v; bi = j; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
        if (ct_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    return CtBest{bv, bi};
}

// grid = n reads, block = 64
__global__ void __launch_bounds__(64) k_cnn_argmax(const float *__restrict__ scores, int n, int Lo, int na, int k,
                                                   long long *__restrict__ apos, long long *__restrict__ ppos)
{
    const int r = blockIdx.x;
    const float *s0 = scores + (size_t)r * 2 * Lo, *s1 = s0 + Lo;
    const int hi = na < Lo ? na : Lo;
    int a = 0;
    if (hi > 0) a = ct_wave_argmax(s0, 0, hi).i;
    int p = 0;
    if (k >= 1) {
        // the masked row is -5 on [0, a): its arg-max is 0 unless something in [a, Lo) beats -5 (or a == 0, or a NaN follows)
        const CtBest b = ct_wave_argmax(s1, a, Lo);
        if (a == 0 || b.v != b.v || b.v > CT_EXCL) p = b.i;
    }
    if (lane_id() == 0) { apos[r] = a; ppos[r] = p; }
}

// lnz[r] = the nearest row x <= r OF THE SAME MINIBATCH with unmasked samples (a_x <= p_x), as an index inside the
// minibatch, or -1.  grid = minibatches (a minibatch is one call of the reference: one flattened array), block = 1024.
__global__ void __launch_bounds__(1024) k_cnn_rowlink(const long long *__restrict__ apos, const long long *__restrict__ ppos, int n_all,
                                                      int mbsize, int32_t *__restrict__ lnz)
{
    __shared__ __attribute__((aligned(16))) int part[1024];
    const int m0 = blockIdx.x * mbsize;
    const int n = min(n_all - m0, mbsize);
    apos += m0; ppos += m0; lnz += m0;
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    int last = -1;
    for (int r = lo; r < hi; r++) if (apos[r] <= ppos[r]) last = r;
    part[t] = last;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { // inclusive prefix maximum
        const int v = t >= o ? part[t - o] : -1;
        __syncthreads();
        if (v > part[t]) part[t] = v;
        __syncthreads();
    }
    int run = t > 0 ? part[t - 1] : -1;
    for (int r = lo; r < hi; r++) { if (apos[r] <= ppos[r]) run = r; lnz[r] = run; }
}

// ---------------------------------------------------------------- the flattened, masked view
struct CtView {
    const float *scores; const long long *apos, *ppos; const int32_t *lnz; int n, Lo;
    __device__ __forceinline__ float at(int r, int j) const
    {
        const long long a = apos[r], p = ppos[r];
        return (j < a || j > p) ? CT_EXCL : scores[((size_t)r * 2 + 1) * Lo + j];
    }
    // nothing interacts across the boundary between rows r and r + 1 (0 <= r < n - 1)
    __device__ bool decoupled(int r) const
    {
        const long long a1 = apos[r + 1], p1 = ppos[r + 1];
        if (!(p1 < a1 || a1 >= 5)) return false; // the next row's stretch starts within 5 samples of the boundary
        const int x = lnz[r];
        if (x < 0) return true;                   // masked back to the start of the array: that run is no peak
        const long long px = ppos[x];
        if (x == r && px > Lo - 6) return false;  // this row's stretch ends within 5 samples of the boundary
        const float vp = scores[((size_t)x * 2 + 1) * Lo + px];
        return vp > CT_EXCL;                      // (else the masked run behind row x may be a peak, or merge with it)
    }
    // last flat position of the run of samples equal to v that contains flat position g (f(g) == v), looking right;
    // `right` = the value behind the run, ok = false when the array ends there
    __device__ long long run_end(long long g, float v, float &right, bool &ok) const
    {
        const long long end = (long long)n * Lo;
        long long e = g;
        for (;;) {
            int r = (int)(e / Lo), j = (int)(e - (long long)r * Lo);
            const long long a = apos[r], p = ppos[r];
            if ((j < a || j > p) && v == CT_EXCL) { // inside a masked run whose value continues the plateau: jump to its end
                if (j < a) e = (long long)r * Lo + (a <= Lo ? a : Lo) - 1;
                else e = (long long)(r + 1) * Lo - 1;
                if (p < a) e = (long long)(r + 1) * Lo - 1; // a row with nothing unmasked
            }
            const long long nx = e + 1;
            if (nx >= end) { ok = false; right = 0.f; return e; }
            r = (int)(nx / Lo); j = (int)(nx - (long long)r * Lo);
            const float fv = at(r, j);
            if (fv == v) { e = nx; continue; }
            ok = true; right = fv;
            return e;
        }
    }
};

template <class SP> static __device__ __forceinline__ uint32_t ct_ld(SP p);
template <> __device__ __forceinline__ uint32_t ct_ld<LDS uint32_t *>(LDS uint32_t *p) { return *p; }
// (state words in global memory are changed by atomics that execute in L2: read them past the L1 as well)
template <> __device__ __forceinline__ uint32_t ct_ld<uint32_t *>(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One chain [r0, r1] by one wave.  pk / mk / pv: scratch lists starting at the chain's first row (half entries per row);
// stw: 2-bit states by ordinal (slot q + 2; 0 none, 3 undecided, 2 kept, 1 removed), zeroed here.
template <class SP>
static __device__ void ct_chain(const CtView &V, int r0, int r1, int k, int32_t *__restrict__ pk, int32_t *__restrict__ mk, float *__restrict__ pv,
                                SP stw, int32_t *__restrict__ cand, int32_t *__restrict__ cnt)
{
    const int ln = lane_id();
    const int Lo = V.Lo, n = V.n;
    const long long last_flat = (long long)n * Lo - 1;
    for (int rr = r0; rr <= r1; rr++)
        for (int q = ln; q < k; q += 64) cand[(size_t)rr * k + q] = 0;
    // 1. local maxima (samples and midpoints of runs of equal samples), in index order
    int npk = 0;
    float carry = __builtin_inff(); // the sample in front of the row (in front of the chain: masked like the chain's first, or none)
    for (int rr = r0; rr <= r1; rr++) {
        const long long a = V.apos[rr], p = V.ppos[rr];
        const float *row = V.scores + ((size_t)rr * 2 + 1) * Lo;
        const bool empty = p < a; // nothing unmasked in this row
        // A masked run holds no left edge of a peak except its first sample.  The run that starts the row:
        if ((empty || a > 0) && carry < CT_EXCL) {
            bool pkf = false; int pos = 0;
            const long long g = (long long)rr * Lo;
            if (ln == 0 && g > 0 && g < last_flat) {
                float right; bool ok;
                const long long e = V.run_end(g, CT_EXCL, right, ok);
                if (ok && right < CT_EXCL) { pkf = true; pos = (int)((g + e) / 2); }
            }
            const unsigned long long m = __ballot(pkf);
            if (pkf) { pk[npk] = pos; pv[npk] = CT_EXCL; }
            npk += __popcll(m);
        }
        if (empty) { carry = CT_EXCL; continue; }
        // the unmasked stretch [a, p] and the first masked sample behind it
        const int lo = (int)a, hi = (int)(p + 1 < Lo - 1 ? p + 1 : Lo - 1);
        if (a > 0) carry = CT_EXCL;
        for (int base = lo; base <= hi; base += 64) {
            const int j = base + ln;
            const bool in = j <= hi;
            const float v = in ? (j > p ? CT_EXCL : row[j]) : 0.f;
            float prev = __shfl_up(v, 1);
            if (ln == 0) prev = carry;
            const int lastl = hi - base < 63 ? hi - base : 63;
            carry = __shfl(v, lastl);
            bool pkf = false; int pos = 0;
            const long long g = (long long)rr * Lo + j;
            if (in && prev < v && g > 0 && g < last_flat) {
                float next; // the sample behind: in this row, or the first of the next row (g < last_flat: it exists)
                if (j + 1 < Lo) next = (j + 1 > p) ? CT_EXCL : row[j + 1];
                else next = V.at(rr + 1, 0);
                if (next < v) { pkf = true; pos = (int)g; }
                else if (next == v) { // a run of equal samples: rare, walked by this lane alone
                    float right; bool ok;
                    const long long e = V.run_end(g, v, right, ok);
                    if (ok && right < v) { pkf = true; pos = (int)((g + e) / 2); }
                }
            }
            const unsigned long long m = __ballot(pkf);
            if (pkf) { const int slot = npk + __popcll(m & ((1ull << ln) - 1ull)); pk[slot] = pos; pv[slot] = v; }
            npk += __popcll(m);
        }
        // carry = the row's last sample in the masked view
        if (p < Lo - 1) carry = CT_EXCL;
    }
    for (int w = ln; w < (npk + 4 + 15) / 16 + 1; w += 64) stw[w] = 0;
    __threadfence_block();
    __syncthreads();
    // 2. neighbourhood masks (ordinals q-2 .. q+2 closer than 5 samples); bits 0,1: q+1, q+2 count as higher; 2,3: q-1, q-2
    for (int base = 0; base < npk; base += 60) {
        const int q = base - 2 + ln;
        const bool valid = q >= 0 && q < npk;
        const int pp = valid ? pk[q] : (q < 0 ? -0x40000000 : 0x7fffffff);
        const float v = valid ? pv[q] : 0.f;
        uint32_t mask = 0;
        const bool out = valid && ln >= 2 && ln < 62; // (lanes 0, 1, 62, 63 only lend their values)
#pragma unroll
        for (int j = 1; j <= 2; j++) {
            const int pf = __shfl_down(pp, j); const float vf = __shfl_down(v, j);
            const int pb = __shfl_up(pp, j);   const float vb = __shfl_up(v, j);
            if (out && (long long)pf - pp <= 4 && vf >= v) mask |= 1u << (j - 1);     // equal heights: the later index first
            if (out && (long long)pp - pb <= 4 && vb > v) mask |= 1u << (2 + j - 1);
        }
        if (out) {
            const uint32_t sl = (uint32_t)(q + 2);
            __hip_atomic_fetch_or(&stw[sl >> 4], (mask ? 3u : 2u) << ((sl & 15u) * 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mk[q] = (int)mask;
        }
    }
    __threadfence_block();
    __syncthreads();
    // 3. fixed point: an undecided maximum is removed by a kept higher neighbour, kept when all higher ones are removed
    for (;;) {
        bool progress = false, left = false;
        for (int base = 0; base < npk; base += 64) {
            const int q = base + ln;
            if (q < npk) {
                const uint32_t sl = (uint32_t)(q + 2);
                const uint32_t st = (ct_ld<SP>(&stw[sl >> 4]) >> ((sl & 15u) * 2u)) & 3u;
                if (st == 3u) {
                    const uint32_t mask = (uint32_t)mk[q];
                    // states of the ordinals q-2 .. q+2 = slots q .. q+4
                    const unsigned long long W = (((unsigned long long)ct_ld<SP>(&stw[(q >> 4) + 1]) << 32) | ct_ld<SP>(&stw[q >> 4])) >> ((q & 15) * 2);
                    bool kept_nb = false, pending = false;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (mask >> b & 1u) {
                            const int f = b < 2 ? 3 + b : 3 - b; // q+1, q+2 -> fields 3, 4 ; q-1, q-2 -> fields 1, 0
                            const uint32_t sn = (uint32_t)(W >> (2 * f)) & 3u;
                            if (sn == 2u) kept_nb = true; else if (sn == 3u) pending = true;
                        }
                    }
                    if (kept_nb) { __hip_atomic_fetch_and(&stw[sl >> 4], ~(2u << ((sl & 15u) * 2u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); progress = true; }
                    else if (!pending) { __hip_atomic_fetch_and(&stw[sl >> 4], ~(1u << ((sl & 15u) * 2u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); progress = true; }
                    else left = true;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        if (!__any(left)) break;
        if (!__any(progress)) break; // (cannot happen: "counts as higher" is a strict total order)
    }
    // 4. per read of the chain: the kept maxima whose position lies in its row; k rounds of arg-max (height descending,
    //    index ascending: np.lexsort((-heights, read_idx)) is stable).  The kept maxima are first packed to the front of the two
    //    lists (in place, in index order: a third to a half of the maxima survive the distance rule, and every round walks them).
    int nk_all = 0;
    for (int base = 0; base < npk; base += 64) {
        const int q = base + ln;
        bool kp = false; int pp = 0; float vv = 0.f;
        if (q < npk) { const uint32_t sl = (uint32_t)(q + 2); pp = pk[q]; vv = pv[q]; kp = ((ct_ld<SP>(&stw[sl >> 4]) >> ((sl & 15u) * 2u)) & 3u) == 2u; }
        const unsigned long long m = __ballot(kp);
        if (kp) { const int at = nk_all + __popcll(m & ((1ull << ln) - 1ull)); pk[at] = pp; pv[at] = vv; } // (at <= q: behind what has been read)
        nk_all += __popcll(m);
    }
    __threadfence_block();
    __syncthreads();
    for (int rr = r0; rr <= r1; rr++) {
        const long long glo = (long long)rr * Lo, ghi = glo + Lo;
        int nkept = 0;
        for (int base = 0; base < nk_all; base += 64) {
            const int q = base + ln;
            bool kp = false;
            if (q < nk_all) { const int pp = pk[q]; kp = pp >= glo && pp < ghi; }
            nkept += __popcll(__ballot(kp));
        }
        const int rounds = nkept < k ? nkept : k;
        float lim_v = __builtin_inff(); int lim_i = -1; // the previous pick: later picks come after it in (height desc, index asc)
        for (int t = 0; t < rounds; t++) {
            float bv = -__builtin_inff(); int bi = 0x7fffffff;
            for (int base = 0; base < nk_all; base += 64) {
                const int q = base + ln;
                if (q < nk_all) {
                    const int i = pk[q];
                    if (i >= glo && i < ghi) {
                        const float v = pv[q];
                        const bool after = (v < lim_v) || (v == lim_v && i > lim_i);
                        if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
                    }
                }
            }
            const float wv = wave_max(bv);
            int ci = (bv == wv) ? bi : 0x7fffffff;
            ci = wave_min(ci);
            if (ln == 0) cand[(size_t)rr * k + t] = (int)(ci - glo);
            lim_v = wv; lim_i = ci;
        }
        if (ln == 0) cnt[rr] = nkept;
    }
}

// grid = n reads, block = 64; dynamic LDS = ((Lo / 2 + 1 + 4) / 16 + 2) words (the states of a single-row chain).
// pk / mk [n, half] int32, pv [n, half] float, stw_g [n, wpr] (states of chains of several rows), half = Lo / 2 + 1;
// cand [n, k] int32 (zero padded), cnt [n] = peaks of the read after the distance rule.  Reads [q mbsize, (q + 1) mbsize)
// form minibatch q = one flattened array of the reference.
__global__ void __launch_bounds__(64) k_cnn_topk(const float *__restrict__ scores, const long long *__restrict__ apos,
                                                 const long long *__restrict__ ppos, const int32_t *__restrict__ lnz, int n_all, int mbsize, int Lo, int k,
                                                 int32_t *__restrict__ pk_all, int32_t *__restrict__ mk_all, float *__restrict__ pv_all,
                                                 uint32_t *__restrict__ stw_g, int wpr,
                                                 int32_t *__restrict__ cand, int32_t *__restrict__ cnt)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ct_raw[];
    const int m0 = (blockIdx.x / mbsize) * mbsize;
    const int n = min(n_all - m0, mbsize);
    const int r = blockIdx.x - m0;
    scores += (size_t)m0 * 2 * Lo; apos += m0; ppos += m0; lnz += m0;
    pk_all += (size_t)m0 * ((size_t)Lo / 2 + 1); mk_all += (size_t)m0 * ((size_t)Lo / 2 + 1); pv_all += (size_t)m0 * ((size_t)Lo / 2 + 1);
    stw_g += (size_t)m0 * wpr; cand += (size_t)m0 * k; cnt += m0;
    CtView V{scores, apos, ppos, lnz, n, Lo};
    if (r > 0 && !V.decoupled(r - 1)) return; // inside a chain: its first row's wave does the work
    int r1 = r;
    while (r1 + 1 < n && !V.decoupled(r1)) r1++;
    const size_t half = (size_t)Lo / 2 + 1;
    int32_t *pk = pk_all + (size_t)r * half, *mk = mk_all + (size_t)r * half;
    float *pv = pv_all + (size_t)r * half;
    if (r1 == r) ct_chain<LDS uint32_t *>(V, r, r1, k, pk, mk, pv, (LDS uint32_t *)ct_raw, cand, cnt);
    else ct_chain<uint32_t *>(V, r, r1, k, pk, mk, pv, stw_g + (size_t)r * wpr, cand, cnt);
}

// ---------------------------------------------------------------- candidate table of the validator
// bounds[i] = (adapter, k candidates) in SAMPLES: index * ds + off, a value equal to off becomes 0 (cnn.py:173-179).
// k > 1: the candidates of the i-th read with peaks OF A MINIBATCH go to its row i (cnn.py:150-158).
// grid = minibatches, block = 1024.
__global__ void __launch_bounds__(1024) k_cnn_bounds(const long long *__restrict__ apos, const long long *__restrict__ ppos,
                                                     const int32_t *__restrict__ cand, const int32_t *__restrict__ cnt, int n_all, int mbsize, int k,
                                                     int ds, int off, int64_t *__restrict__ bounds)
{
    __shared__ __attribute__((aligned(16))) int part[1024];
    const int t = threadIdx.x;
    const int kk = k < 1 ? 1 : k; // columns behind the adapter
    const int m0 = blockIdx.x * mbsize;
    const int n = min(n_all - m0, mbsize);
    apos += m0; ppos += m0; cand += (size_t)m0 * kk; cnt += m0; bounds += (size_t)m0 * (1 + kk);
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    auto conv = [&](long long idx) { const long long v = idx * ds + off; return v == off ? 0ll : v; };
    for (int r = lo; r < hi; r++) {
        bounds[(size_t)r * (1 + kk)] = conv(apos[r]);
        if (k <= 1) bounds[(size_t)r * (1 + kk) + 1] = conv(k == 1 ? ppos[r] : 0);
        else for (int c = 0; c < k; c++) bounds[(size_t)r * (1 + kk) + 1 + c] = conv(0);
    }
    if (k <= 1) return;
    int mine = 0;
    for (int r = lo; r < hi; r++) mine += cnt[r] > 0;
    part[t] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { // inclusive prefix sum
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int rank = part[t] - mine;
    __syncthreads(); // (every row's zero fill above is done before the scattered writes below)
    for (int r = lo; r < hi; r++) {
        if (cnt[r] > 0) {
            for (int c = 0; c < k; c++) bounds[(size_t)rank * (1 + kk) + 1 + c] = conv(c < cnt[r] ? cand[(size_t)r * k + c] : 0);
            rank++;
        }
    }
}
