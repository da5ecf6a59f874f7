// trace_api.h -- the reference's native module as a batched device API (SURVEY 8(b) "Native", 8(f) rank 4):
// `c_llr_trace`, `c_llr_trace_gains`, `_gains` of adapted/detect/_c_llr.pyx:67-236 with stride and both early-stopping forms
// (`_gains_w_early_stop` :91-122, `_gains_w_polya_early_stop` :125-173), for n_reads float64 signals of one call.
//
// The product path (adp_detect_llr) has its own specialised passes (llr_stream.h: float32 pooled input, checkpointed sums,
// fixed offsets); this is the general form behind `adp_c_llr_trace` -- any (start, end, offsets, stride) per read -- for callers
// that bind the library where the reference binds its Cython module (llr.py:18).
//
// Early stopping is sequential in the reference (`break` inside the loop over i).  Every check only reads gains at indices
// BELOW the i it is made at, and a break leaves exactly those computed, so the checks can be evaluated on the full trace:
// the wave computes all gains of the read, then finds the first i whose check fires and zeroes the trace from there on --
// the same array the sequential loop leaves.  The checks' `np.diff(gains[i-w:i:stride]).mean()` is restated with numpy's
// summation order (add.reduce: pairwise, 8 accumulators per leaf of <= 128, <= 8192 elements per inner loop) and Python's
// slice rule for a negative start.
#pragma once
#include "common.h"
#include "log_cr.h"

struct TraceArgs { int min_obs, border_trim, stride, a_es, a_w, a_s, p_es, p_w, p_s; };

// np.cumsum(raw), np.cumsum(raw * raw): one wave per read; tiles of 512 samples through LDS (coalesced loads and stores),
// the two chains run by lane 0.  grid = n_reads, block = 64.
__global__ void __launch_bounds__(64) k_trace_cumsum(const double *__restrict__ raw, const int32_t *__restrict__ len, int L, int n_reads,
                                                     double *__restrict__ c, double *__restrict__ c2)
{
    constexpr int TILE = 512;
    __shared__ __attribute__((aligned(16))) double sx[TILE], sc[TILE], sq[TILE];
    const int r = blockIdx.x, ln = threadIdx.x;
    const int n = len[r];
    const double *x = raw + (size_t)r * L;
    double *co = c + (size_t)r * L, *c2o = c2 + (size_t)r * L;
    double a = 0.0, b = 0.0;
    for (int tb = 0; tb < n; tb += TILE) {
        const int cnt = min(TILE, n - tb);
        for (int k = ln; k < cnt; k += 64) sx[k] = x[tb + k];
        __syncthreads();
        if (ln == 0) {
#pragma unroll 8
            for (int k = 0; k < cnt; k++) {
                const double v = sx[k], q = v * v;
                if (tb + k == 0) { a = v; b = q; } // (the first element as it is: -0.0 stays -0.0)
                else { a += v; b += q; }
                sc[k] = a; sq[k] = b;
            }
        }
        __syncthreads();
        for (int k = ln; k < cnt; k += 64) { co[tb + k] = sc[k]; c2o[tb + k] = sq[k]; }
        __syncthreads();
    }
    for (int k = n + ln; k < L; k += 64) { co[k] = 0.0; c2o[k] = 0.0; }
}

// numpy's add.reduce over d[k] = g[lo + (k + 1) s] - g[lo + k s], k in [0, nd): pairwise recursion with an explicit stack
// (numpy/core/src/umath/loops_utils.h.src pairwise_sum; the inner loop sees <= 8192 elements at a time)
static __device__ double tr_pairwise_diff(const double *g, long lo, long s, long off, long nd)
{
    auto d = [&](long k) { return g[lo + (off + k + 1) * s] - g[lo + (off + k) * s]; };
    auto leaf = [&](long o, long n) {
        if (n < 8) {
            double res = 0.0;
            for (long i = 0; i < n; i++) res += d(o + i);
            return res;
        }
        double r0 = d(o), r1 = d(o + 1), r2 = d(o + 2), r3 = d(o + 3), r4 = d(o + 4), r5 = d(o + 5), r6 = d(o + 6), r7 = d(o + 7);
        long i;
        for (i = 8; i < n - (n % 8); i += 8) {
            r0 += d(o + i); r1 += d(o + i + 1); r2 += d(o + i + 2); r3 += d(o + i + 3);
            r4 += d(o + i + 4); r5 += d(o + i + 5); r6 += d(o + i + 6); r7 += d(o + i + 7);
        }
        double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < n; i++) res += d(o + i);
        return res;
    };
    if (nd <= 128) return leaf(0, nd);
    // post-order walk: (offset, length, phase) nodes, the results of finished subtrees on a value stack
    long so[16], sn[16];
    int sph[16], sp = 0, rs = 0;
    double val[16];
    so[0] = 0; sn[0] = nd; sph[0] = 0; sp = 1;
    while (sp > 0) {
        sp--;
        const long o = so[sp], n = sn[sp];
        if (sph[sp]) { const double rr = val[--rs], ll = val[--rs]; val[rs++] = ll + rr; continue; }
        if (n <= 128) { val[rs++] = leaf(o, n); continue; }
        long n2 = n / 2;
        n2 -= n2 % 8;
        so[sp] = o; sn[sp] = n; sph[sp] = 1; sp++;             // combine, after both halves
        so[sp] = o + n2; sn[sp] = n - n2; sph[sp] = 0; sp++;   // right half (popped second)
        so[sp] = o; sn[sp] = n2; sph[sp] = 0; sp++;            // left half (popped first)
    }
    return val[0];
}

// np.diff(g[lo:i:stride]).mean(); NaN for fewer than two elements (the mean of an empty array)
static __device__ double tr_diff_mean(const double *g, long n, long lo, long i, long s)
{
    if (lo < 0) { lo += n; if (lo < 0) lo = 0; } // Python's rule for a negative slice start
    long cnt = 0;
    if (i > lo) cnt = (i - lo + s - 1) / s;
    if (cnt < 2) return __builtin_nan("");
    const long nd = cnt - 1;
    double t = 0.0;
    for (long o = 0; o < nd; o += 8192) t += tr_pairwise_diff(g, lo, s, o, nd - o < 8192 ? nd - o : 8192);
    return t / (double)nd;
}

// the gains of one read, then the early-stopping rules.  grid = n_reads, block = 64 (one wave; __syncthreads orders its
// global stores before the loads of the checks).
__global__ void __launch_bounds__(64) k_trace_gains(const double *__restrict__ c, const double *__restrict__ c2, const int32_t *__restrict__ len,
                                                    const int32_t *__restrict__ start_, const int32_t *__restrict__ end_, int L, TraceArgs a,
                                                    double *__restrict__ gain)
{
    __shared__ __attribute__((aligned(16))) double lt_[3 * LOGCR_N];
    const int r = blockIdx.x, ln = threadIdx.x;
    for (int i = ln; i < 3 * LOGCR_N; i += 64) lt_[i] = g_logcr_table[i];
    const long n = len[r], start = start_[r], end = end_[r];
    double *g = gain + (size_t)r * L;
    for (int i = ln; i < L; i += 64) g[i] = 0.0;
    __syncthreads();
    const LDS double *lt = (const LDS double *)lt_;
    auto flog = [&](double v) { return log_cr_impl(v, lt, [](double u) { return log(u); }); };
    const double *cr = c + (size_t)r * L, *c2r = c2 + (size_t)r * L;
    // var_c (_c_llr.pyx:23-37): 0 for an empty segment; the sums before `lo` are 0 for lo = 0 (x - 0.0 = x: the same bits as
    // the reference's branch without the subtraction)
    auto var_c = [&](long lo, long hi) {
        if (lo == hi) return 0.0;
        const double d = (double)(hi - lo);
        const double cl = lo > 0 ? cr[lo - 1] : 0.0, c2l = lo > 0 ? c2r[lo - 1] : 0.0;
        const double mu = (cr[hi - 1] - cl) / d;
        return (c2r[hi - 1] - c2l) / d - mu * mu;
    };
    const long s0 = start + a.min_obs, stop = end - a.border_trim, s = a.stride;
    if (n > 0 && s0 < stop) {
        const double vs = (double)(end - start) * flog(var_c(start, end));
        for (long i = s0 + (long)ln * s; i < stop; i += 64 * s) {
            const double h = (double)(i - start) * flog(var_c(start, i));
            const double t = (double)(end - i) * flog(var_c(i, end));
            g[i] = vs - (h + t);
        }
    }
    if (!(a.p_es > 0 || a.a_es > 0) || !(n > 0 && s0 < stop)) return;
    __syncthreads();
    // first i of the adapter rule: i = s0 + j a_s >= s0 + a_w with mean(diff(g[i - a_w : i : s])) < 0
    long ia = -1;
    {
        const long j0 = (a.a_w + a.a_s - 1) / a.a_s; // (j = 0 only passes for a_w = 0: an empty window, a NaN mean, no stop)
        for (long jb = j0; s0 + jb * a.a_s < stop && ia < 0; jb += 64) {
            const long i = s0 + (jb + ln) * a.a_s;
            bool hit = false;
            if (i < stop) hit = tr_diff_mean(g, n, i - a.a_w, i, s) < 0.0;
            const unsigned long long mk = __ballot(hit);
            if (mk) ia = s0 + (jb + (__ffsll((long long)mk) - 1)) * a.a_s;
        }
    }
    long ib = -1; // the i the loop breaks at
    if (a.p_es > 0) {
        // from the iteration that found the adapter on, every i: mean(diff(g[i - p_w : i : s])) > 0 ends the loop
        if (ia >= 0) {
            for (long kb = ia; kb < stop && ib < 0; kb += 64 * s) {
                const long i = kb + (long)ln * s;
                bool hit = false;
                if (i < stop) hit = tr_diff_mean(g, n, i - a.p_w, i, s) > 0.0;
                const unsigned long long mk = __ballot(hit);
                if (mk) ib = kb + (long)(__ffsll((long long)mk) - 1) * s;
            }
        }
    } else ib = ia;
    if (ib < 0) return;
    __syncthreads(); // (uniform: ib is the same in every lane)
    for (long i = ib + ln; i < n; i += 64) g[i] = 0.0;
}
