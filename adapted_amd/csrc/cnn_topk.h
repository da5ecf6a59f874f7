// cnn_topk.h -- C3: the k best poly(A) candidates per read from the CNN's channel-1 scores.
//
// reference adapted/detect/cnn.py:136-160: mask the scores before the adapter position and after the best
// poly(A) position to -5, run scipy.signal.find_peaks(distance=5) on the FLATTENED [n * Lo] array, group the
// peaks by read, order each group by descending height (ties: ascending index) and keep the first k.
//
// One wave per read.  Between the unmasked stretches [a_r, p_r] of consecutive reads the flattened array is a
// plateau of -5, so the reads only meet in corner cases; those, and everything else whose outcome depends on
// scipy's handling of exact ties, are DETECTED here and reported (flag != 0): the host then runs the reference's
// numpy/scipy formulation for that batch.  Conditions reported per read:
//   * a read without a masked prefix whose scores all lie below the mask level: the masked plateau after it
//     could be a peak;
//   * two equal neighbouring scores inside the stretch (a plateau), or two maxima of equal height within the
//     minimum distance (scipy's tie order comes from an unstable argsort);
//   * the stretches of two consecutive reads come within the minimum distance of each other across the row
//     boundary (a <= 4 and p_prev >= Lo - 5).
// Otherwise: strict local maxima of the stretch (its end points see the mask level or the neighbouring read's
// edge sample), minimum distance 5 as the fixed point of "kept iff no kept higher maximum within 4 samples" on the
// ordinals of the maxima (at most two per side lie that close), then k rounds of arg-max over the kept ones.
#pragma once
#include "common.h"

#define CT_EXCL (-5.0f)
#define CT_FLAG_EMPTY 1
#define CT_FLAG_PLATEAU 2
#define CT_FLAG_TIE 4
#define CT_FLAG_BOUNDARY 8

// masked channel-1 value of flat position (r, j); j may be -1 or Lo (neighbouring read).  ok = false beyond the array.
static __device__ __forceinline__ float ct_val(const float *__restrict__ scores, const long long *__restrict__ apos,
                                               const long long *__restrict__ ppos, int n, int Lo, int r, int j, bool &ok)
{
    if (j < 0) { r--; j += Lo; }
    else if (j >= Lo) { r++; j -= Lo; }
    ok = r >= 0 && r < n;
    if (!ok) return 0.f;
    const long long a = apos[r], p = ppos[r];
    if (j < a || j > p) return CT_EXCL;
    return scores[((size_t)r * 2 + 1) * Lo + j];
}

// grid = n reads, block = 64; dynamic LDS = ((Lo / 2 + 1 + 4) / 16 + 2) words (2-bit states by ordinal, offset 2)
// pk: per-read scratch [n, 2, Lo/2 + 1] (positions, neighbour masks), pv [n, Lo/2 + 1] (heights);
// cand [n, k] int32 (zero padded); cnt [n] = kept maxima
__global__ void __launch_bounds__(64) k_cnn_topk(const float *__restrict__ scores, const long long *__restrict__ apos,
                                                 const long long *__restrict__ ppos, int n, int Lo, int k,
                                                 int32_t *__restrict__ pk_all, float *__restrict__ pv_all,
                                                 int32_t *__restrict__ cand, int32_t *__restrict__ cnt, int32_t *__restrict__ flag)
{
    extern __shared__ uint32_t ct_raw[];
    LDS uint32_t *stw = (LDS uint32_t *)ct_raw; // ordinal q -> slot q + 2; 0 none, 3 undecided, 2 kept, 1 removed
    const int r = blockIdx.x;
    const int ln = lane_id();
    const int half = Lo / 2 + 1;
    int32_t *pk = pk_all + (size_t)r * 2 * half;
    int32_t *mk = pk + half;
    float *pv = pv_all + (size_t)r * half;
    const float *row = scores + ((size_t)r * 2 + 1) * Lo;
    const int a = (int)apos[r], p = (int)ppos[r];
    int fl = 0;
    for (int q = ln; q < k; q += 64) cand[(size_t)r * k + q] = 0;
    // A run of masked samples (it may span several reads) is bounded on the left by row_x[p_x] of some read x; that is
    // the maximum of x's masked row, so it lies below the mask level only if x has no masked prefix (a_x = 0) and all its
    // scores are below -5: only then can a masked plateau be a peak.
    if (a == 0 && p >= a && row[p] < CT_EXCL) fl |= CT_FLAG_EMPTY;
    if (r > 0 && a <= 4 && ppos[r - 1] >= Lo - 5) fl |= CT_FLAG_BOUNDARY;
    if (fl) { if (ln == 0) { atomicOr(flag, fl); cnt[r] = 0; } return; }
    if (p < a) { if (ln == 0) cnt[r] = 0; return; } // nothing unmasked: no peak in this read
    // neighbours of the stretch's end points
    bool okl, okr;
    const float vleft = ct_val(scores, apos, ppos, n, Lo, r, a - 1, okl);
    const float vright = ct_val(scores, apos, ppos, n, Lo, r, p + 1, okr);
    // 1. strict local maxima of [a, p], in index order
    int npk = 0;
    float carry = vleft; // value before the tile
    for (int base = a; base <= p; base += 64) {
        const int i = base + ln;
        const float v = (i <= p) ? row[i] : 0.f;
        float prev = __shfl_up(v, 1);
        if (ln == 0) prev = carry;
        float next = __shfl_down(v, 1);
        if (ln == 63 || i == p) next = (i == p) ? vright : ((i + 1 <= p) ? row[i + 1] : 0.f);
        carry = __shfl(v, 63);
        bool pkf = false;
        if (i <= p) {
            const bool has_l = (i > a) || okl, has_r = (i < p) || okr; // (the ends of the flattened array are never maxima)
            if ((i < p || okr) && v == next) fl |= CT_FLAG_PLATEAU; // (also a plateau running into the mask or the next read)
            if (i == a && okl && prev == v) fl |= CT_FLAG_PLATEAU;
            pkf = has_l && has_r && prev < v && next < v;
        }
        const unsigned long long m = __ballot(pkf);
        if (pkf) { const int slot = npk + __popcll(m & ((1ull << ln) - 1ull)); pk[slot] = i; pv[slot] = v; }
        npk += __popcll(m);
    }
    for (int w = ln; w < (npk + 4 + 15) / 16 + 1; w += 64) stw[w] = 0;
    __syncthreads();
    // 2. neighbourhood masks (ordinals q-2 .. q+2; |dp| <= 4), maxima without a higher neighbour are kept at once
    //    mask bits: 0,1 = q+1, q+2 higher; 2,3 = q-1, q-2 higher
    for (int base = 0; base < npk; base += 60) {
        const int q = base - 2 + ln;
        const bool valid = q >= 0 && q < npk;
        const int pp = valid ? pk[q] : (q < 0 ? -0x40000000 : 0x40000000);
        const float v = valid ? pv[q] : 0.f;
        uint32_t mask = 0;
        const bool out = valid && ln >= 2 && ln < 62; // (lanes 0, 1, 62, 63 only lend their values)
#pragma unroll
        for (int j = 1; j <= 2; j++) {
            const int pf = __shfl_down(pp, j); const float vf = __shfl_down(v, j);
            const int pb = __shfl_up(pp, j);   const float vb = __shfl_up(v, j);
            if (out && pf - pp <= 4) { if (vf == v) fl |= CT_FLAG_TIE; if (vf > v) mask |= 1u << (j - 1); }
            if (out && pp - pb <= 4) { if (vb == v) fl |= CT_FLAG_TIE; if (vb > v) mask |= 1u << (2 + j - 1); }
        }
        if (out) {
            const uint32_t sl = (uint32_t)(q + 2);
            __hip_atomic_fetch_or(&stw[sl >> 4], (mask ? 3u : 2u) << ((sl & 15u) * 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            mk[q] = (int)mask;
        }
    }
    __syncthreads();
    // 3. fixed point: an undecided maximum is removed by a kept higher neighbour, kept when all higher ones are removed
    for (;;) {
        bool progress = false, left = false;
        for (int base = 0; base < npk; base += 64) {
            const int q = base + ln;
            if (q < npk) {
                const uint32_t sl = (uint32_t)(q + 2);
                const uint32_t st = (stw[sl >> 4] >> ((sl & 15u) * 2u)) & 3u;
                if (st == 3u) {
                    const uint32_t mask = (uint32_t)mk[q];
                    // states of the ordinals q-2 .. q+2 = slots q .. q+4
                    const unsigned long long W = (((unsigned long long)stw[(q >> 4) + 1] << 32) | stw[q >> 4]) >> ((q & 15) * 2);
                    bool kept_nb = false, pending = false;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (mask >> b & 1u) {
                            const int f = b < 2 ? 3 + b : 3 - b; // q+1, q+2 -> fields 3, 4 ; q-1, q-2 -> fields 1, 0
                            const uint32_t sn = (uint32_t)(W >> (2 * f)) & 3u;
                            if (sn == 2u) kept_nb = true; else if (sn == 3u) pending = true;
                        }
                    }
                    if (kept_nb) { __hip_atomic_fetch_and(&stw[sl >> 4], ~(2u << ((sl & 15u) * 2u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); progress = true; }
                    else if (!pending) { __hip_atomic_fetch_and(&stw[sl >> 4], ~(1u << ((sl & 15u) * 2u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); progress = true; }
                    else left = true;
                }
            }
            __syncthreads();
        }
        if (!__any(left)) break;
        if (!__any(progress)) { fl |= CT_FLAG_TIE; break; } // (cannot happen without ties)
    }
    // 4. kept maxima: count, then k rounds of arg-max (height descending, index ascending)
    int nkept = 0;
    for (int base = 0; base < npk; base += 64) {
        const int q = base + ln;
        bool kp = false;
        if (q < npk) { const uint32_t sl = (uint32_t)(q + 2); kp = ((stw[sl >> 4] >> ((sl & 15u) * 2u)) & 3u) == 2u; }
        nkept += __popcll(__ballot(kp));
    }
    const int rounds = nkept < k ? nkept : k;
    float lim_v = __builtin_inff(); int lim_i = -1; // the previous pick: later picks come after it in (height desc, index asc)
    for (int t = 0; t < rounds; t++) {
        float bv = -__builtin_inff(); int bi = 0x7fffffff;
        for (int base = 0; base < npk; base += 64) {
            const int q = base + ln;
            if (q < npk) {
                const uint32_t sl = (uint32_t)(q + 2);
                if (((stw[sl >> 4] >> ((sl & 15u) * 2u)) & 3u) == 2u) {
                    const float v = pv[q]; const int i = pk[q];
                    const bool after = (v < lim_v) || (v == lim_v && i > lim_i);
                    if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
                }
            }
        }
        const float wv = wave_max(bv);
        int ci = (bv == wv) ? bi : 0x7fffffff;
        ci = wave_min(ci);
        if (ln == 0) cand[(size_t)r * k + t] = ci;
        lim_v = wv; lim_i = ci;
    }
    if (__any(fl != 0)) { int all = fl; for (int o = 32; o > 0; o >>= 1) all |= __shfl_xor(all, o); if (ln == 0) atomicOr(flag, all); }
    if (ln == 0) cnt[r] = nkept;
}
