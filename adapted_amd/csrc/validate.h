// validate.h -- V1 validate_boundaries and everything under it, one wave per read.
//
//   V1 validate_boundaries            reference adapted/detect/combined.py:358-631
//   V2 find_open_pores                adapted/detect/anomalies.py:15-35
//   V3 real_range_check               adapted/detect/real_range.py:33-63
//   V4 mean_var_shift_polyA_check     adapted/detect/mvs.py:45-158   (bottleneck move_mean /
//      move_var: float32 streaming recurrences, strictly sequential -> run by single lanes)
//   S1 calc_partition_stats           adapted/partition/signal_partitions.py:81-96
//   K2 start-peak decorations         adapted/detect/combined.py:333-347
//
// The kernel is a persistent grid of waves ("slots"); each slot owns scratch for the moving
// mean / variance series of one poly(A) candidate.  All branches are per-read decisions and
// therefore wave-uniform.
#pragma once
#include <type_traits>
#ifndef VAL_MULTI_RANKS
#define VAL_MULTI_RANKS 1 // k_validate: the percentile pairs (and the median beside them) of one slice by ONE multi-rank selection (wave_stats.h)
#endif
#include "common.h"
#include "wave_stats.h"
#include "llr_stream.h"
#include "block_stats.h"
#include "cand_stats.h"

template <class SIG>
struct ValidateInT {
    SIG sig;                   // [n_reads, m] (float32 pA, or int16 ADC + calibration)
    const int32_t *full_len;   // [n_reads]
    const int64_t *bounds;     // [n_reads, 1 + kmax]: adapter_end, candidates (0 terminates)
    const int8_t *topk_none;   // [n_reads] 1 <=> polya_end_topk is None ; may be nullptr (all given)
    int kmax;
    int n_reads, m, mbsize;
    const MbState *mbs;        // may be nullptr (no minibatch gating)
    float *scratch;            // [slots, 2, scratch_stride]
    int scratch_stride;
    const float *series;       // [n_reads, 2, series_cap] moving mean / var up to the largest candidate (k_mvs_series) or nullptr
    const int8_t *have_series; // [n_reads] 1: `series` holds the read's moving mean / variance.  INVARIANT every series producer keeps
                               // (k_mvs_series, k_mvs_series_wave, k_mvs_series_pipe): have = 0 for a slice with a NaN in it -- mvs_check
                               // skips its own NaN scan for slices whose series came from a series kernel (tests/test_gpu_variants.py
                               // plants NaN holes early in a slice and inside its last window and reads this flag back)
    int series_cap;
    const CandStat *cstat;     // [n_reads, kmax] the candidates' order statistics (k_cand_stats) or nullptr
    int32_t *op_arena;         // whole open_pores lists of the reads with more than ADP_MAX_OPEN_PORES entries
    unsigned int *op_used;     // entries handed out (may exceed op_cap: the host then grows the arena and repeats the kernel)
    unsigned int op_cap;
};

static __device__ __forceinline__ bool in_range_d(double v, double lo, double hi) { return lo <= v && v <= hi; }
static __device__ __forceinline__ bool range_empty(const double *r)
{
    return __builtin_isinf(r[0]) && r[0] < 0 && __builtin_isinf(r[1]) && r[1] > 0;
}

struct RowW {
    adp_row *row;
    unsigned long long present;
    __device__ void set(int c, double v) { if (lane_id() == 0) row->col[c] = v; present |= 1ull << c; }
};

// The two bottleneck recurrences below are strictly sequential chains in one lane.  Their loads do not depend on
// the chain, so they are issued eight steps at a time, a block AHEAD of the arithmetic: otherwise every step
// would wait for its own round trip to memory.
#define BN_BLK 8
// eight consecutive samples as two 16-byte accesses (gfx950 needs dword alignment only): with one LANE per read every
// access of the wave touches 64 different rows, so the number of memory instructions is what the kernel costs
typedef float bn_f4 __attribute__((ext_vector_type(4), aligned(4)));
template <class X>
static __device__ __forceinline__ void bn_load8(float (&dst)[BN_BLK], X a, int i, int n, int back)
{
    if (i + BN_BLK <= n) {
        const float4 v0 = a.f4uc(i - back), v1 = a.f4uc(i - back + 4);
        dst[0] = v0.x; dst[1] = v0.y; dst[2] = v0.z; dst[3] = v0.w; dst[4] = v1.x; dst[5] = v1.y; dst[6] = v1.z; dst[7] = v1.w;
    } else {
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) dst[j] = a.at_or(i + j - back, i + j < n, 0.f);
    }
}
static __device__ __forceinline__ void bn_store8(GLB float *out, int o, const float (&res)[BN_BLK], int i, int n)
{
    if (i + BN_BLK <= n) {
        const bn_f4 v0 = {res[0], res[1], res[2], res[3]}, v1 = {res[4], res[5], res[6], res[7]};
        *reinterpret_cast<GLB bn_f4 *>(out + o) = v0;
        *reinterpret_cast<GLB bn_f4 *>(out + o + 4) = v1;
    } else {
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) if (i + j < n) out[o + j] = res[j];
    }
}

// bottleneck.move_mean(a, window=w) for i >= w-1 (float32, NaN-free input) -- single lane
template <class X>
static __device__ __noinline__ void bn_move_mean(X a, int n, int w, float *out_)
{
    GLB float *out = (GLB float *)out_;
    float asum = 0.f;
    for (int i0 = 0; i0 < w; i0 += BN_BLK) {
        float v[BN_BLK];
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) v[j] = a.at_or(i0 + j, i0 + j < w, 0.f);
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) if (i0 + j < w) asum += v[j];
    }
    out[0] = asum / (float)w;
    const float inv = (float)(1.0 / (double)w);
    float cn[BN_BLK], co[BN_BLK], nn[BN_BLK], no[BN_BLK];
    bn_load8(cn, a, w, n, 0);
    bn_load8(co, a, w, n, w);
    for (int i0 = w; i0 < n; i0 += BN_BLK) {
        bn_load8(nn, a, i0 + BN_BLK, n, 0);
        bn_load8(no, a, i0 + BN_BLK, n, w);
        float res[BN_BLK];
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) {
            asum += cn[j] - co[j];
            res[j] = asum * inv;
        }
        bn_store8(out, i0 - w + 1, res, i0, n);
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) { cn[j] = nn[j]; co[j] = no[j]; }
    }
}

// bottleneck.move_var(a, window=w, ddof=0) for i >= w-1 -- single lane
template <class X>
static __device__ __noinline__ void bn_move_var(X a, int n, int w, float *out_)
{
    GLB float *out = (GLB float *)out_;
    float amean = 0.f, assqdm = 0.f;
    int count = 0;
    for (int i0 = 0; i0 < w; i0 += BN_BLK) {
        float v[BN_BLK];
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) v[j] = a.at_or(i0 + j, i0 + j < w, 0.f);
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) {
            if (i0 + j < w) {
                const float ai = v[j];
                count++;
                float delta = ai - amean;
                amean += delta / (float)count;
                assqdm += delta * (ai - amean);
            }
        }
    }
    if (assqdm < 0) assqdm = 0;
    out[0] = assqdm / (float)count;
    const float ddof_inv = (float)(1.0 / (double)count), count_inv = ddof_inv;
    float cn[BN_BLK], co[BN_BLK], nn[BN_BLK], no[BN_BLK];
    bn_load8(cn, a, w, n, 0);
    bn_load8(co, a, w, n, w);
    for (int i0 = w; i0 < n; i0 += BN_BLK) {
        bn_load8(nn, a, i0 + BN_BLK, n, 0);
        bn_load8(no, a, i0 + BN_BLK, n, w);
        float res[BN_BLK] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) {
            if (i0 + j < n) { // (the padding of the last block must not touch the state)
                float ai = cn[j], aold = co[j];
                float delta = ai - aold;
                aold -= amean;
                amean += delta * count_inv;
                ai -= amean;
                assqdm += (ai + aold) * delta;
                if (assqdm < 0) assqdm = 0;
                res[j] = assqdm * ddof_inv;
            }
        }
        bn_store8(out, i0 - w + 1, res, i0, n);
#pragma unroll
        for (int j = 0; j < BN_BLK; j++) { cn[j] = nn[j]; co[j] = no[j]; }
    }
}

// The same two recurrences element by element, for k_validate's rare long-window path over int16 rows: the register
// budget of anything k_validate can reach counts against its 6 waves per SIMD, and the eight-sample pipelines above need
// 103 registers once each sample is also converted.  Same operations, same order.
template <class X>
static __device__ __noinline__ void bn_move_slim(X a, int n, int w, float *out_, bool var)
{
    GLB float *out = (GLB float *)out_;
    float amean = 0.f, assqdm = 0.f, asum = 0.f;
    int count = 0;
    for (int i = 0; i < w; i++) {
        const float ai = a[i];
        if (var) { count++; const float delta = ai - amean; amean += delta / (float)count; assqdm += delta * (ai - amean); }
        else asum += ai;
    }
    const float inv = (float)(1.0 / (double)w);
    if (var) { if (assqdm < 0) assqdm = 0; out[0] = assqdm / (float)count; }
    else out[0] = asum / (float)w;
    for (int i = w; i < n; i++) {
        float ai = a[i], aold = a[i - w];
        if (var) {
            const float delta = ai - aold;
            aold -= amean; amean += delta * inv; ai -= amean;
            assqdm += (ai + aold) * delta;
            if (assqdm < 0) assqdm = 0;
            out[i - w + 1] = assqdm * inv;
        } else { asum += ai - aold; out[i - w + 1] = asum * inv; }
    }
}

// Both recurrences for one read inside a wave (k_validate: candidates whose series k_mvs_series did not prepare).
// The wave copies the slice into LDS chunk by chunk with coalesced loads (plus the window's worth of history the
// sliding steps subtract); lane 0 then runs move_var and lane 1 move_mean over the chunk from LDS, so the chains
// wait for memory once per chunk instead of once per few samples.  Same arithmetic as bn_move_var / bn_move_mean.
#define MV_CHUNK 768
#define MV_HIST (WS_STAGE_FLOATS - MV_CHUNK) // longest window this path serves
template <class X>
static __device__ void wave_move_series(X x, int n, int wv, int wm, bool do_var, bool do_mean, float *svar_, float *smean_,
                                        LDS WaveScratch *ws)
{
    GLB float *svar = (GLB float *)svar_;
    GLB float *smean = (GLB float *)smean_;
    const int ln = lane_id();
    LDS float *buf = ws->stage;
    float amean = 0.f, assqdm = 0.f, asum = 0.f; // lane 0: variance state; lane 1: sum
    int count = 0;
    const float inv_v = (float)(1.0 / (double)wv), inv_m = (float)(1.0 / (double)wm);
    for (int i0 = 0; i0 < n; i0 += MV_CHUNK) {
        const int lo = i0 >= MV_HIST ? i0 - MV_HIST : 0; // buf[k] = x[lo + k]
        const int hi = min(n, i0 + MV_CHUNK);
        ws_sync();
        for (int k = ln; k < hi - lo; k += 64) buf[k] = x[lo + k];
        ws_sync();
        if (ln == 0 && do_var) {
            int i = i0;
            for (; i >= wv && i + 8 <= hi; i += 8) { // sliding steps, eight at a time: the LDS reads go first
                float an[8], ao[8], res[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { an[j] = buf[i + j - lo]; ao[j] = buf[i + j - wv - lo]; }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float ai = an[j], aold = ao[j];
                    float delta = ai - aold;
                    aold -= amean;
                    amean += delta * inv_v;
                    ai -= amean;
                    assqdm += (ai + aold) * delta;
                    if (assqdm < 0) assqdm = 0;
                    res[j] = assqdm * inv_v;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) svar[i + j - wv + 1] = res[j];
            }
            for (; i < hi; i++) {
                float ai = buf[i - lo];
                if (i < wv) {
                    count++;
                    float delta = ai - amean;
                    amean += delta / (float)count;
                    assqdm += delta * (ai - amean);
                    if (i == wv - 1) { if (assqdm < 0) assqdm = 0; svar[0] = assqdm / (float)count; }
                } else {
                    float aold = buf[i - wv - lo];
                    float delta = ai - aold;
                    aold -= amean;
                    amean += delta * inv_v;
                    ai -= amean;
                    assqdm += (ai + aold) * delta;
                    if (assqdm < 0) assqdm = 0;
                    svar[i - wv + 1] = assqdm * inv_v;
                }
            }
        }
        if (ln == 1 && do_mean) {
            int i = i0;
            for (; i >= wm && i + 8 <= hi; i += 8) {
                float an[8], ao[8], res[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { an[j] = buf[i + j - lo]; ao[j] = buf[i + j - wm - lo]; }
#pragma unroll
                for (int j = 0; j < 8; j++) { asum += an[j] - ao[j]; res[j] = asum * inv_m; }
#pragma unroll
                for (int j = 0; j < 8; j++) smean[i + j - wm + 1] = res[j];
            }
            for (; i < hi; i++) {
                float ai = buf[i - lo];
                if (i < wm) {
                    asum += ai;
                    if (i == wm - 1) smean[0] = asum / (float)wm;
                } else {
                    asum += ai - buf[i - wm - lo];
                    smean[i - wm + 1] = asum * inv_m;
                }
            }
        }
    }
    ws_sync();
}

// calc_partition_stats -> (start, len, mean, std, med, mad)
template <class X>
static __device__ void partition_stats(X sig, int S, long long start, long long end, RowW &rw, int c_start,
                                       int c_len, LDS WaveScratch *ws, bool have_medmad, float med_in, float mad_in)
{
    rw.set(c_start, (double)start);
    if (end <= start) return;
    rw.set(c_len, (double)(end - start));
    long long a = start < S ? start : S, b = end < S ? end : S;
    int n = (int)(b - a);
    float mean, sd, med, mad;
    if (n <= 0) { mean = sd = med = mad = __builtin_nanf(""); }
    else {
        const X x = sig + a;
        float var = wave_np_var(x, n, ws, &mean);
        sd = sqrtf(var);
        if (have_medmad) { med = med_in; mad = mad_in; }
        else { med = wave_median(x, n, 0, 0.f, ws); mad = wave_median(x, n, 1, med, ws); }
    }
    rw.set(c_len + 1, (double)mean);
    rw.set(c_len + 2, (double)sd);
    rw.set(c_len + 3, (double)med);
    rw.set(c_len + 4, (double)mad);
}

// ---------------------------------------------------------------- NaN samples inside a slice
// The loaders never produce them (calibrated int16), the hot kernels assume there are none -- their recurrences would
// carry a NaN forever --, but the operators accept any float32 array and bottleneck counts NaN samples out of the window
// (a window with fewer than `window` valid samples yields NaN; later windows are clean again), so a slice with a NaN takes
// these single-lane restatements of bottleneck/src/move_template.c instead, and np.nanmedian over what they return.
static __device__ __noinline__ void bn_move_mean_nan(RowF32 a, int n, int w, float *out_)
{
    GLB float *out = (GLB float *)out_;
    const float qnan = __builtin_nanf("");
    float asum = 0.f;
    int count = 0;
    for (int i = 0; i < w; i++) { const float ai = a[i]; if (ai == ai) { asum += ai; count++; } }
    out[0] = count >= w ? asum / (float)count : qnan;
    float inv = (float)(1.0 / (double)count);
    for (int i = w; i < n; i++) {
        const float ai = a[i], aold = a[i - w];
        if (ai == ai) {
            if (aold == aold) asum += ai - aold;
            else { asum += ai; count++; inv = (float)(1.0 / (double)count); }
        } else if (aold == aold) { asum -= aold; count--; inv = (float)(1.0 / (double)count); }
        out[i - w + 1] = count >= w ? asum * inv : qnan;
    }
}

static __device__ __noinline__ void bn_move_var_nan(RowF32 a, int n, int w, float *out_)
{
    GLB float *out = (GLB float *)out_;
    const float qnan = __builtin_nanf("");
    float amean = 0.f, assqdm = 0.f;
    int count = 0;
    for (int i = 0; i < w; i++) {
        const float ai = a[i];
        if (ai == ai) { count++; const float delta = ai - amean; amean += delta / (float)count; assqdm += delta * (ai - amean); }
    }
    if (count >= w) { if (assqdm < 0) assqdm = 0; out[0] = assqdm / (float)count; }
    else out[0] = qnan;
    float inv = (float)(1.0 / (double)count);
    for (int i = w; i < n; i++) {
        float ai = a[i], aold = a[i - w];
        if (ai == ai) {
            if (aold == aold) {
                const float delta = ai - aold;
                aold -= amean; amean += delta * inv; ai -= amean;
                assqdm += (ai + aold) * delta;
            } else {
                count++; inv = (float)(1.0 / (double)count);
                const float delta = ai - amean;
                amean += delta * inv;
                assqdm += delta * (ai - amean);
            }
        } else if (aold == aold) {
            count--; inv = (float)(1.0 / (double)count);
            if (count > 0) { const float delta = aold - amean; amean -= delta * inv; assqdm -= delta * (aold - amean); }
            else { amean = 0.f; assqdm = 0.f; }
        }
        if (count >= w) { if (assqdm < 0) assqdm = 0; out[i - w + 1] = assqdm * inv; }
        else out[i - w + 1] = qnan;
    }
}

// any NaN in x[0..n)?  (the same answer in every lane; int16 rows have none)
template <class X>
static __device__ __forceinline__ bool wave_has_nan(X x, int n)
{
    if constexpr (std::is_same<X, RowI16>::value) return false;
    else {
        bool bad = false;
        for (int i = lane_id(); i < n; i += 64) { const float v = x[i]; bad |= v != v; }
        return __any(bad);
    }
}

// np.nanmedian(s[0..n))
static __device__ __noinline__ float wave_nanmedian(const float *s, int n, LDS WaveScratch *ws)
{
    int cnt = 0;
    for (int i = lane_id(); i < n; i += 64) cnt += s[i] == s[i];
    cnt = wave_sum(cnt);
    if (cnt <= 0) return __builtin_nanf("");
    float vk, vkm1;
    wave_select2_skipnan(s, n, cnt / 2, 0, 0.f, ws, vk, vkm1);
    return (cnt & 1) ? vk : (vkm1 + vk) / 2.0f;
}

struct MvsOut { int ok, vec_fail, exc; double mean, var, med, lrange, shift; };

// shift_cache: the median shift across the adapter end is the same for every candidate of a read (value, flag)
template <class X>
static __device__ __noinline__ MvsOut mvs_check(X sig, int S, long long a_e, long long p_e, const adp_cfg &cfg, double pr0,
                                   double pr1, LDS WaveScratch *ws, float *scr_mean, float *scr_var, LDS SegCache *sc,
                                   const float *pre_mean, const float *pre_var, const CandStat *cst, float &shift_val, bool &shift_have,
                                   bool series_clean = false)
{
    MvsOut o; o.ok = 0; o.vec_fail = 31; o.exc = 0; o.mean = o.var = o.med = o.lrange = o.shift = 0.0;
    if (p_e == 0 || a_e == 0 || p_e < a_e || p_e - a_e <= 2) return o;
    if ((long long)S < a_e + cfg.median_shift_window) return o;
    const int a = (int)(a_e < S ? a_e : S), b = (int)(p_e < S ? p_e : S);
    const int n = b - a;
    const X x = sig + a;
    const bool wvar = !(p_e - a_e <= cfg.pA_var_window + 2), wmean = !(p_e - a_e <= cfg.pA_mean_window + 2);
    if ((wvar && (cfg.pA_var_window > n || cfg.pA_var_window < 1)) || (wmean && (cfg.pA_mean_window > n || cfg.pA_mean_window < 1))) {
        o.exc = ADP_F_EXC_MOVE_WINDOW; return o;
    }
    float fvar, fmean, fmed;
    double lrange;
    if (cst && cst->ready && wvar && wmean) { // all five order statistics of this candidate came out of the shared sweeps
        fvar = cst->fvar; fmean = cst->fmean; fmed = cst->fmed; lrange = cst->q85 - cst->q15;
    } else {
    // the two sequential recurrences run side by side in lanes 0 and 1
    __syncthreads();
    // (series prepared by the series kernels exist only for slices without a NaN: a NaN stays in the recurrences to their end and withdraws them)
    const bool nanx = (pre_mean && series_clean) ? false : wave_has_nan(x, n);
    if (nanx) {
        if constexpr (!std::is_same<X, RowI16>::value) {
            if (lane_id() == 0 && wvar) bn_move_var_nan(x, n, cfg.pA_var_window, scr_var);
            if (lane_id() == 1 && wmean) bn_move_mean_nan(x, n, cfg.pA_mean_window, scr_mean);
        }
        __threadfence_block();
    } else if (pre_mean) { scr_mean = const_cast<float *>(pre_mean); scr_var = const_cast<float *>(pre_var); }
    else {
        if (cfg.pA_var_window <= MV_HIST && cfg.pA_mean_window <= MV_HIST) {
            wave_move_series(x, n, cfg.pA_var_window, cfg.pA_mean_window, wvar, wmean, scr_var, scr_mean, ws);
        } else {
            if constexpr (std::is_same<X, RowI16>::value) {
                if (lane_id() == 0 && wvar) bn_move_slim(x, n, cfg.pA_var_window, scr_var, true);
                if (lane_id() == 1 && wmean) bn_move_slim(x, n, cfg.pA_mean_window, scr_mean, false);
            } else {
                if (lane_id() == 0 && wvar) bn_move_var(x, n, cfg.pA_var_window, scr_var);
                if (lane_id() == 1 && wmean) bn_move_mean(x, n, cfg.pA_mean_window, scr_mean);
            }
        }
        __threadfence_block();
    }
    __syncthreads();
    if (sc && lane_id() == 0) sc->n = 0; // the scratch series were rewritten: drop any mirror of them
    __syncthreads();
    if (wvar) fvar = nanx ? wave_nanmedian(scr_var, n - cfg.pA_var_window + 1, ws) : wave_median(scr_var, n - cfg.pA_var_window + 1, 0, 0.f, ws, sc);
    else fvar = wave_np_var(x, n, ws, nullptr);
    if (wmean) fmean = nanx ? wave_nanmedian(scr_mean, n - cfg.pA_mean_window + 1, ws) : wave_median(scr_mean, n - cfg.pA_mean_window + 1, 0, 0.f, ws, sc);
    else fmean = wave_np_mean(x, n, ws);
    if (VAL_MULTI_RANKS && n > 0) wave_median_local_range(x, n, ws, fmed, lrange); // (median and both percentiles in the passes of one selection)
    else {
    fmed = wave_median(x, n, 0, 0.f, ws, sc);
    lrange = (n > 0) ? wave_percentile(x, n, 85.0, ws, sc) - wave_percentile(x, n, 15.0, ws, sc) : (double)__builtin_nanf("");
    }
    }
    if (!shift_have) {
        long long r1 = a_e + cfg.median_shift_window; if (r1 > S) r1 = S;
        long long l0 = a_e - cfg.median_shift_window; if (l0 < 0) l0 = 0;
        shift_val = wave_median(sig + a, (int)(r1 - a), 0, 0.f, ws, sc) - wave_median(sig + l0, (int)(a - l0), 0, 0.f, ws, sc);
        shift_have = true;
    }
    const float shift = shift_val;
    o.mean = (double)fmean; o.var = (double)fvar; o.med = (double)fmed; o.lrange = lrange; o.shift = (double)shift;
    int f = 0;
    if (!in_range_d(o.mean, pr0, pr1)) f |= 1;
    if (!in_range_d(o.var, cfg.pA_var_range[0], cfg.pA_var_range[1])) f |= 2;
    if (!in_range_d(o.med, cfg.polyA_med_range[0], cfg.polyA_med_range[1])) f |= 4;
    if (!in_range_d(o.lrange, cfg.polyA_local_range[0], cfg.polyA_local_range[1])) f |= 8;
    if (!in_range_d(o.shift, cfg.median_shift_range[0], cfg.median_shift_range[1])) f |= 16;
    o.vec_fail = f; o.ok = (f == 0);
    return o;
}

// mvs_detect_overwrite: mean_var_shift_polyA_detect_at_loc(signal, loc = adapter_end, less_signal_ok = False)
// (reference adapted/detect/mvs.py:181-338).  The moving mean / variance of signal[loc - offset, loc + search_window)
// (offset = the longer window) are scanned for the first position with both in range; the array comparisons are
// float32 against the bounds cast to float32 (numpy 1.x value-based casting, utils.py:26), the scalar checks float64.
struct MvsLoc { int ok, exc; long long idx; double mean, var, med, lrange, shift; };

static __device__ __forceinline__ bool in_range_f32(float v, double lo, double hi) { return (float)lo <= v && v <= (float)hi; }

template <class X>
static __device__ __noinline__ MvsLoc mvs_detect_at_loc(X sig, int S, long long loc, const adp_cfg &cfg, double pr0, double pr1,
                                                        LDS WaveScratch *ws, float *scr_mean, float *scr_var, LDS SegCache *sc,
                                                        float med_before_loc)
{
    MvsLoc o; o.ok = 0; o.exc = 0; o.idx = 0; o.mean = o.var = o.med = o.lrange = o.shift = 0.0;
    const int wm = cfg.pA_mean_window, wv = cfg.pA_var_window;
    const int offset = wm > wv ? wm : wv;
    const int tailw = cfg.median_shift_window > cfg.polyA_window ? cfg.median_shift_window : cfg.polyA_window;
    if ((long long)S < loc + cfg.search_window + tailw) return o; // not enough signal after loc (:216-231)
    if (loc < offset) return o;                                   // ... or before it (:234-247)
    const int n = offset + cfg.search_window;
    const X x = sig + (loc - offset);
    if (wm < 1 || wv < 1) { o.exc = ADP_F_EXC_MOVE_WINDOW; return o; } // (windows longer than the slice cannot occur)
    __syncthreads();
    // the series hold the outputs from index window-1 on (the first window-1 are NaN in bottleneck: never in range)
    if (wave_has_nan(x, n)) {
        if constexpr (!std::is_same<X, RowI16>::value) {
            if (lane_id() == 0) bn_move_var_nan(x, n, wv, scr_var);
            if (lane_id() == 1) bn_move_mean_nan(x, n, wm, scr_mean);
        }
    } else if (wv <= MV_HIST && wm <= MV_HIST) wave_move_series(x, n, wv, wm, true, true, scr_var, scr_mean, ws);
    else {
        if constexpr (std::is_same<X, RowI16>::value) {
            if (lane_id() == 0) bn_move_slim(x, n, wv, scr_var, true);
            if (lane_id() == 1) bn_move_slim(x, n, wm, scr_mean, false);
        } else {
            if (lane_id() == 0) bn_move_var(x, n, wv, scr_var);
            if (lane_id() == 1) bn_move_mean(x, n, wm, scr_mean);
        }
    }
    __threadfence_block();
    __syncthreads();
    if (sc && lane_id() == 0) sc->n = 0;
    __syncthreads();
    int idx = 0;
    for (int base = offset - 1; base < n; base += 64) {
        const int i = base + lane_id();
        const bool hit = i < n && in_range_f32(scr_mean[i - wm + 1], pr0, pr1) &&
                         in_range_f32(scr_var[i - wv + 1], cfg.pA_var_range[0], cfg.pA_var_range[1]);
        const unsigned long long mk = __ballot(hit);
        if (mk) { idx = base + __ffsll((long long)mk) - 1; break; }
    }
    const int at = idx > 0 ? idx : 2 * offset; // not found: the values one window lag behind loc (set_config: 2*offset < n)
    o.mean = (double)scr_mean[at - wm + 1];
    o.var = (double)scr_var[at - wv + 1];
    long long ix = idx > 0 ? (long long)idx + loc - offset : 0;
    o.idx = ix;
    const long long loc_ = loc > ix ? loc : ix;
    long long e1 = loc_ + cfg.polyA_window; if (e1 > S) e1 = S;
    long long e2 = loc_ + cfg.median_shift_window; if (e2 > S) e2 = S;
    const X y = sig + loc_;
    float fmed;
    if (VAL_MULTI_RANKS && e1 - loc_ > 0) wave_median_local_range(y, (int)(e1 - loc_), ws, fmed, o.lrange);
    else {
        fmed = wave_median(y, (int)(e1 - loc_), 0, 0.f, ws, sc);
        o.lrange = wave_percentile(y, (int)(e1 - loc_), 85.0, ws, sc) - wave_percentile(y, (int)(e1 - loc_), 15.0, ws, sc);
    }
    o.med = (double)fmed;
    const float before = (loc_ == loc) ? med_before_loc : wave_median(sig, (int)loc_, 0, 0.f, ws, sc);
    o.shift = (double)(wave_median(y, (int)(e2 - loc_), 0, 0.f, ws, sc) - before);
    o.ok = idx > 0 && in_range_d(o.med, cfg.polyA_med_range[0], cfg.polyA_med_range[1]) &&
           in_range_d(o.lrange, cfg.polyA_local_range[0], cfg.polyA_local_range[1]) &&
           in_range_d(o.shift, cfg.median_shift_range[0], cfg.median_shift_range[1]);
    return o;
}

// ---------------------------------------------------------------- V4 moving mean / variance series, ahead of time
// The bottleneck recurrences are strictly sequential per read; inside k_validate they would occupy one
// lane of a wave.  They only depend on (adapter_end, first poly(A) candidate), which are known before
// validation starts, so this kernel runs them for candidate 0 with one LANE per read (64 reads per wave).
#define MVS_CAP 8192
template <class SIG>
__global__ void __launch_bounds__(64) k_mvs_series(SIG sigs, const int32_t *__restrict__ full_len, int n_reads,
                                                   int m, const int64_t *__restrict__ bounds, int kmax, adp_cfg cfg,
                                                   float *__restrict__ series, int cap, int8_t *__restrict__ have)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_reads) return;
    have[r] = 0;
    const long long fl = full_len[r];
    const int S = (int)(fl < m ? fl : m);
    // the recurrences are causal: the series of the candidate with the LARGEST poly(A) end holds those of the
    // others (same start, the adapter end) as prefixes -- one series serves all candidates of the read
    const long long a_e = bounds[(size_t)r * (1 + kmax)];
    long long p_e = 0;
    for (int c = 0; c < kmax; c++) { const long long pc = bounds[(size_t)r * (1 + kmax) + 1 + c]; if (pc == 0) break; if (pc > p_e) p_e = pc; }
    if (p_e == 0 || a_e == 0 || p_e < a_e || p_e - a_e <= 2) return;
    if ((long long)S < a_e + cfg.median_shift_window) return;
    const int a = (int)(a_e < S ? a_e : S), b = (int)(p_e < S ? p_e : S);
    const int n = b - a;
    const bool wvar = !(p_e - a_e <= cfg.pA_var_window + 2), wmean = !(p_e - a_e <= cfg.pA_mean_window + 2);
    if ((wvar && (cfg.pA_var_window > n || cfg.pA_var_window < 1)) || (wmean && (cfg.pA_mean_window > n || cfg.pA_mean_window < 1))) return;
    if (n > cap) return;
    const typename SIG::Row x = sigs.row(r, m) + a;
    float *smean = series + (size_t)r * 2 * cap, *svar = smean + cap;
    if (wvar) bn_move_var(x, n, cfg.pA_var_window, svar);
    if (wmean) bn_move_mean(x, n, cfg.pA_mean_window, smean);
    // a NaN sample stays in these recurrences to the end: such a read is k_validate's business (bn_move_*_nan)
    bool clean = true;
    if (wvar) { const float t = svar[n - cfg.pA_var_window]; clean &= t == t; }
    if (wmean) { const float t = smean[n - cfg.pA_mean_window]; clean &= t == t; }
    have[r] = clean ? 1 : 0;
}

// The same series for LONG slices (the CNN path's candidates at wide windows: up to the whole preload).  The chains are
// strictly sequential float32 recurrences (same operations, same order as bn_move_var / bn_move_mean), so a read cannot be
// spread over lanes -- but several READS can share a wave: a workgroup of two waves takes MS_G reads, wave 0 runs their
// moving variances (lane = read), wave 1 their moving means, one vector instruction advancing all MS_G chains.  Each wave
// keeps an LDS ring per read, filled with coalesced loads (64 lanes fetch 64 consecutive samples of one read at a time,
// requested a chunk ahead of the chains) and read by the chains with 16-byte accesses (rows 4 x odd floats apart:
// conflict-free); the results go back through LDS and leave as coalesced stores.  The kernel is bound by the LATENCY of the
// longest chain, not by throughput (the same time for 1000 and 4000 reads).  History at the 200 k window, per 4000 reads:
// one lane per read straight from global memory (k_mvs_series) 38 ms -- the chains waited on scattered loads; one chain per
// WAVE from LDS 23 ms -- eight waves per SIMD, one lane in 64 working, bound by instruction issue; MS_G chains per wave with
// the chunk loads waited for one by one 23 ms; with the next chunk's loads in flight during the chains 12.9 ms; round 2,
// late: the loads made unconditional (the compiler had put their wait BEFORE the chains, where two paths joined), the stores
// a chunk late (one counter for loads and stores), the variance steps hand-scheduled (ms_var4), the LDS reads a group
// ahead, the ring writes unconditional: 10.9 ms.  Ablation (ADP_ABLATE-style switches, since removed): chains 7 ms --
// 45 instructions per four steps at the 4-cycle issue rate of a single wave --, LDS copies and loop 3, loads + stores 1.4.
#define MS_CHUNK 64
#define MS_HIST 320   // longest window served
#define MS_G 16       // reads per wave (= chains advanced by one vector instruction)

// ring of RB floats per read (RB = a power of two >= window + MS_CHUNK): sample i of the slice lives at ring[i & (RB - 1)];
// rows are RB + 4 floats apart (4 x odd: the 16-byte accesses of the 16 chain lanes fall into different bank groups)
static __device__ __forceinline__ int ms_ring(int w) { int rb = 128; while (rb < w + MS_CHUNK) rb <<= 1; return rb; }

// Four sliding steps of bottleneck's move_var (same operations, same order as bn_move_var) as ONE hand-scheduled block.
// One wave per SIMD runs these chains, and a vector instruction that needs its predecessor's result issues ~8 cycles behind
// it instead of 4: the compiler's order (step after step, nearly every instruction waiting for the one before, plus packed
// subtractions that cost more moves than they save: 68 instructions, ~480 cycles per four steps) made the kernel 2.3x slower
// than the carried dependencies require.  Here the four steps are interleaved so that (nearly) no instruction follows its
// producer directly; only the two carried chains remain: the mean (one add per step) and the sum of squares (add, compare,
// select per step -- `if (assqdm < 0) assqdm = 0` keeps a NaN, so no v_max).  The compare writes vcc; gfx950 wants two
// wait states before a v_cndmask reads it: two other instructions stand between them everywhere but at the end (s_nop 1).
//   per step: delta = ai - aold; aold -= amean; amean += delta * inv; ai -= amean; assqdm += (ai + aold) * delta;
//             if (assqdm < 0) assqdm = 0; out = assqdm * inv
// registers: D_j delta_j, later out_j; E_j delta_j * inv, later the mean after step j; P_j aold_j - mean, later the clamped
// sum after step j; Q_j ai_j - mean, the product, the unclamped sum.
static __device__ __forceinline__ void ms_var4(const float (&a)[4], const float (&o)[4], float inv, float &amean, float &assqdm, float (&res)[4])
{
    float d0, d1, d2, d3, e0, e1, e2, e3, p0, p1, p2, p3, q0, q1, q2, q3;
    asm("v_sub_f32_e32 %[d0], %[a0], %[o0]\n\t"
        "v_sub_f32_e32 %[d1], %[a1], %[o1]\n\t"
        "v_mul_f32_e32 %[e0], %[d0], %[inv]\n\t"
        "v_sub_f32_e32 %[d2], %[a2], %[o2]\n\t"
        "v_sub_f32_e32 %[p0], %[o0], %[m]\n\t"
        "v_add_f32_e32 %[e0], %[m], %[e0]\n\t"
        "v_mul_f32_e32 %[e1], %[d1], %[inv]\n\t"
        "v_sub_f32_e32 %[q0], %[a0], %[e0]\n\t"
        "v_sub_f32_e32 %[d3], %[a3], %[o3]\n\t"
        "v_add_f32_e32 %[q0], %[q0], %[p0]\n\t"
        "v_add_f32_e32 %[e1], %[e0], %[e1]\n\t"
        "v_mul_f32_e32 %[q0], %[q0], %[d0]\n\t"
        "v_sub_f32_e32 %[p1], %[o1], %[e0]\n\t"
        "v_add_f32_e32 %[q0], %[s], %[q0]\n\t"
        "v_sub_f32_e32 %[q1], %[a1], %[e1]\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %[q0]\n\t"
        "v_mul_f32_e32 %[e2], %[d2], %[inv]\n\t"
        "v_add_f32_e32 %[q1], %[q1], %[p1]\n\t"
        "v_cndmask_b32_e32 %[p0], 0, %[q0], vcc\n\t"
        "v_mul_f32_e32 %[q1], %[q1], %[d1]\n\t"
        "v_add_f32_e32 %[e2], %[e1], %[e2]\n\t"
        "v_add_f32_e32 %[q1], %[p0], %[q1]\n\t"
        "v_mul_f32_e32 %[e3], %[d3], %[inv]\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %[q1]\n\t"
        "v_sub_f32_e32 %[q2], %[a2], %[e2]\n\t"
        "v_sub_f32_e32 %[p2], %[o2], %[e1]\n\t"
        "v_cndmask_b32_e32 %[p1], 0, %[q1], vcc\n\t"
        "v_add_f32_e32 %[e3], %[e2], %[e3]\n\t"
        "v_add_f32_e32 %[q2], %[q2], %[p2]\n\t"
        "v_mul_f32_e32 %[d0], %[p0], %[inv]\n\t"
        "v_mul_f32_e32 %[q2], %[q2], %[d2]\n\t"
        "v_sub_f32_e32 %[q3], %[a3], %[e3]\n\t"
        "v_add_f32_e32 %[q2], %[p1], %[q2]\n\t"
        "v_sub_f32_e32 %[p3], %[o3], %[e2]\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %[q2]\n\t"
        "v_add_f32_e32 %[q3], %[q3], %[p3]\n\t"
        "v_mul_f32_e32 %[d1], %[p1], %[inv]\n\t"
        "v_cndmask_b32_e32 %[p2], 0, %[q2], vcc\n\t"
        "v_mul_f32_e32 %[q3], %[q3], %[d3]\n\t"
        "v_mul_f32_e32 %[d2], %[p2], %[inv]\n\t"
        "v_add_f32_e32 %[q3], %[p2], %[q3]\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %[q3]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e32 %[p3], 0, %[q3], vcc\n\t"
        "v_mul_f32_e32 %[d3], %[p3], %[inv]"
        : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3),
          [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3)
        : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]),
          [inv] "v"(inv), [m] "v"(amean), [s] "v"(assqdm)
        : "vcc");
    res[0] = d0; res[1] = d1; res[2] = d2; res[3] = d3;
    amean = e3;
    assqdm = p3;
}

template <bool VAR>
static __device__ void ms_chains(const float *__restrict__ sigs, int m, const LDS int32_t *a_of, const LDS int32_t *n_of,
                                 const LDS int32_t *rid_of, int w, float *__restrict__ series, int cap, LDS float *buf, LDS float *out,
                                 int8_t *__restrict__ have)
{
    const int ln = lane_id();
    const int RB = ms_ring(w), S = RB + 4, SO = MS_CHUNK + 4, MASK = RB - 1;
    const int g = ln & (MS_G - 1);
    const int r = rid_of[g];                          // (a read of the launch even where the group has none: its length is 0 then)
    const int n = n_of[g];                            // every lane knows the length of read (lane % MS_G)
    const bool chain = ln < MS_G && n > 0;
    int nmax = n;
#pragma unroll
    for (int o = MS_G / 2; o > 0; o >>= 1) { const int t = __shfl_xor(nmax, o); nmax = t > nmax ? t : nmax; }
    float amean = 0.f, assqdm = 0.f, asum = 0.f;
    int count = 0;
    const float inv = (float)(1.0 / (double)w);
    LDS float *mybuf = buf + g * S;
    const bool vec = (w & 3) == 0;
    // lengths and slice starts of the wave's reads in registers (wave-uniform values)
    int nq[MS_G];
    const GLB float *xq[MS_G];
#pragma unroll
    for (int q = 0; q < MS_G; q++) {
        nq[q] = n_of[q];
        xq[q] = (const GLB float *)sigs + (size_t)rid_of[q] * m + a_of[q];
    }
    // the samples of the NEXT chunk are requested before the chains of the current one run (one load per read, all in
    // flight together; waiting for them one by one cost 8 us per chunk) and land in LDS after it.  The loads are
    // UNCONDITIONAL (index clamped into the slice; what lies behind a slice's end is not copied to LDS): behind a
    // condition the compiler joins the two paths right after the load -- s_waitcnt vmcnt(0) BEFORE the chains, i.e. a full
    // memory round trip (2.5 us, 16 streams with a page each) per chunk of 1.6 us of arithmetic
    float pre[MS_G];
    auto fetch = [&](int i0) {
#pragma unroll
        for (int q = 0; q < MS_G; q++) { const int i = i0 + ln, last = nq[q] > 0 ? nq[q] - 1 : 0; pre[q] = xq[q][i < last ? i : last]; }
    };
    // out[q][i - i0] is read q's series value at index i - w + 1 (defined from i = w - 1 on): coalesced stores per read,
    // issued a chunk LATE, from the other half of the out buffer, right behind the next chunk's loads: gfx950 counts loads
    // and stores in one counter (vmcnt), and the wait for a chunk's samples at the top of the loop would otherwise also wait
    // for the stores issued just before it -- a memory round trip per chunk
    auto store_chunk = [&](int i0, const LDS float *o) {
#pragma unroll
        for (int q = 0; q < MS_G; q++) {
            GLB float *sp = (GLB float *)series + (size_t)rid_of[q] * 2 * cap + (VAR ? cap : 0);
            const int i = i0 + ln;
            if (i < nq[q] && i >= w - 1) sp[i - w + 1] = o[q * SO + ln];
        }
    };
    fetch(0);
    int par = 0;
    for (int i0 = 0; i0 < nmax; i0 += MS_CHUNK, par ^= 1) {
        ws_sync();
#pragma unroll
        for (int q = 0; q < MS_G; q++) buf[q * S + ((i0 + ln) & MASK)] = pre[q]; // (behind a slice's end: cells no chain reads)
        ws_sync();
        fetch(i0 + MS_CHUNK); // (also behind the last chunk: clamped indices, values not used -- no second path to join)
        if (i0 > 0) store_chunk(i0 - MS_CHUNK, out + (par ^ 1) * MS_G * SO);
        LDS float *myout = out + par * MS_G * SO + g * SO;
        if (chain && i0 < n) {
            const int hi = min(n, i0 + MS_CHUNK);
            int i = i0;
            for (; i < hi && i < w; i++) { // the window fills: the reference's first phase
                const float ai = mybuf[i & MASK];
                if (VAR) {
                    count++;
                    const float delta = ai - amean;
                    amean += delta / (float)count;
                    assqdm += delta * (ai - amean);
                    if (i == w - 1) { if (assqdm < 0) assqdm = 0; myout[i - i0] = assqdm / (float)count; }
                } else {
                    asum += ai;
                    if (i == w - 1) myout[i - i0] = asum / (float)w;
                }
            }
            for (; i < hi && (!vec || (i & 3)); i++) { // single sliding steps up to a multiple of 4 (all of them for odd windows)
                float ai = mybuf[i & MASK], aold = mybuf[(i - w) & MASK];
                if (VAR) {
                    const float delta = ai - aold;
                    aold -= amean; amean += delta * inv; ai -= amean;
                    assqdm += (ai + aold) * delta;
                    if (assqdm < 0) assqdm = 0;
                    myout[i - i0] = assqdm * inv;
                } else { asum += ai - aold; myout[i - i0] = asum * inv; }
            }
            // four sliding steps on 16-byte LDS accesses; the NEXT group's samples are requested before this group's
            // arithmetic (ms_var4 wants all eight values at its first instructions: an LDS round trip per group otherwise)
            if (i + 4 <= hi) {
                adp_v4f a4n = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[i & MASK]);
                adp_v4f o4n = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[(i - w) & MASK]);
                for (; i + 4 <= hi; i += 4) {
                    const float an[4] = {a4n.x, a4n.y, a4n.z, a4n.w}, ao[4] = {o4n.x, o4n.y, o4n.z, o4n.w};
                    const int inx = (i + 8 <= hi) ? i + 4 : i; // (the last group reads its own samples again: no second path)
                    a4n = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[inx & MASK]);
                    o4n = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[(inx - w) & MASK]);
                    float res[4];
                    if (VAR) ms_var4(an, ao, inv, amean, assqdm, res);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; j++) { asum += an[j] - ao[j]; res[j] = asum * inv; }
                    }
                    const adp_v4f r4 = {res[0], res[1], res[2], res[3]};
                    *reinterpret_cast<LDS adp_v4f *>(&myout[i - i0]) = r4;
                }
            }
            for (; i < hi; i++) {
                float ai = mybuf[i & MASK], aold = mybuf[(i - w) & MASK];
                if (VAR) {
                    const float delta = ai - aold;
                    aold -= amean; amean += delta * inv; ai -= amean;
                    assqdm += (ai + aold) * delta;
                    if (assqdm < 0) assqdm = 0;
                    myout[i - i0] = assqdm * inv;
                } else { asum += ai - aold; myout[i - i0] = asum * inv; }
            }
        }
    }
    ws_sync();
    if (nmax > 0) store_chunk((nmax - 1) / MS_CHUNK * MS_CHUNK, out + (par ^ 1) * MS_G * SO);
    // a NaN sample stays in the chain to its end: such a read's series are withdrawn (k_validate: bn_move_*_nan)
    if (chain && n > 0 && (VAR ? assqdm != assqdm : asum != asum)) have[r] = 0;
}

// ---- the plan of the long-slice series: which reads get them, from where, how long -- and an ORDER by falling length.
// A wave advances its MS_G chains in lockstep until the longest ends, and the launch ends with its longest workgroup: with the
// reads as they come a wave's time is the MAXIMUM of 16 lengths (2.2 x their mean on the CNN path's candidates at the 200 k
// window) and the long waves start whenever their turn comes.  Ordered by length (a counting sort over MS_NBKT length classes:
// k_series_plan counts, k_series_order places), a wave's chains end together and the longest start first.
#define MS_NBKT 256
static __device__ __forceinline__ int ms_bucket(int n, int cap) // 0: no series; longer slices, larger classes
{
    if (n <= 0) return 0;
    const long long b = 1 + (long long)n * (MS_NBKT - 2) / ((long long)cap + 1);
    return (int)(b < MS_NBKT - 1 ? b : MS_NBKT - 1);
}

// grid = ceil(n_reads / 256), block = 256.  a_of / n_of: slice start and length (0: no series for this read); cnt [MS_NBKT] zeroed
__global__ void __launch_bounds__(256) k_series_plan(const int32_t *__restrict__ full_len, int n_reads, int m, const int64_t *__restrict__ bounds,
                                                     int kmax, adp_cfg cfg, int cap, int8_t *__restrict__ have, int32_t *__restrict__ a_of,
                                                     int32_t *__restrict__ n_of, uint32_t *__restrict__ cnt)
{
    __shared__ __attribute__((aligned(16))) uint32_t hs_[MS_NBKT];
    LDS uint32_t *hs = (LDS uint32_t *)hs_;
    hs[threadIdx.x] = 0;
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_reads) {
        int a = 0, n = 0;
        const long long fl = full_len[r];
        const int S = (int)(fl < m ? fl : m);
        const long long a_e = bounds[(size_t)r * (1 + kmax)];
        long long p_e = 0;
        for (int c = 0; c < kmax; c++) { const long long pc = bounds[(size_t)r * (1 + kmax) + 1 + c]; if (pc == 0) break; if (pc > p_e) p_e = pc; }
        bool ok = !(p_e == 0 || a_e == 0 || p_e < a_e || p_e - a_e <= 2) && !((long long)S < a_e + cfg.median_shift_window);
        a = (int)(a_e < S ? a_e : S);
        const int b = (int)(p_e < S ? p_e : S);
        n = b - a;
        // (both series or none: a slice shorter than a window + 2 is k_validate's business, as is a window that does not fit)
        if (ok && (p_e - a_e <= cfg.pA_var_window + 2 || p_e - a_e <= cfg.pA_mean_window + 2)) ok = false;
        if (ok && (cfg.pA_var_window > n || cfg.pA_var_window < 1 || cfg.pA_mean_window > n || cfg.pA_mean_window < 1)) ok = false;
        if (ok && (n > cap || cfg.pA_var_window > MS_HIST || cfg.pA_mean_window > MS_HIST)) ok = false;
        if (!ok) { n = 0; a = 0; } // (a = 0: the unconditional prefetch of an unused read stays inside its row)
        have[r] = ok ? 1 : 0;
        a_of[r] = a; n_of[r] = n;
        __hip_atomic_fetch_add(&hs[ms_bucket(n, cap)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    const uint32_t c = hs[threadIdx.x];
    if (c) atomicAdd(cnt + threadIdx.x, c); // (one device-scope atomic per class and block: ~100 blocks)
}

// grid = ceil(n_reads / 256), block = 256.  perm: the reads by falling length class; cursor [MS_NBKT] zeroed
__global__ void __launch_bounds__(256) k_series_order(int n_reads, int cap, const int32_t *__restrict__ n_of, const uint32_t *__restrict__ cnt,
                                                      uint32_t *__restrict__ cursor, int32_t *__restrict__ perm)
{
    __shared__ __attribute__((aligned(16))) uint32_t start_[MS_NBKT];
    LDS uint32_t *start = (LDS uint32_t *)start_;
    {   // reads in longer classes come first: start[b] = sum of cnt[b' > b]
        const int b = threadIdx.x;
        uint32_t v = cnt[b];
        start[b] = v;
        __syncthreads();
        uint32_t acc = 0;
        for (int t = b + 1; t < MS_NBKT; t++) acc += start[t];
        __syncthreads();
        start[b] = acc;
        __syncthreads();
    }
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_reads) {
        const int b = ms_bucket(n_of[r], cap);
        const uint32_t pos = start[b] + atomicAdd(cursor + b, 1u);
        perm[pos] = r;
    }
}

// grid = ceil(n_reads / MS_G); block = 128 (wave 0: MS_G moving variances, wave 1: MS_G moving means); dynamic LDS
__global__ void __launch_bounds__(128) k_mvs_series_wave(const float *__restrict__ sigs, int n_reads, int m, const int32_t *__restrict__ a_plan,
                                                         const int32_t *__restrict__ n_plan, const int32_t *__restrict__ perm, adp_cfg cfg,
                                                         float *__restrict__ series, int cap, int8_t *__restrict__ have)
{
    extern __shared__ __attribute__((aligned(16))) float ms_raw[];
    __shared__ __attribute__((aligned(16))) int32_t a_of_[MS_G], n_of_[MS_G], rid_of_[MS_G];
    LDS int32_t *a_of = (LDS int32_t *)a_of_, *n_of = (LDS int32_t *)n_of_, *rid_of = (LDS int32_t *)rid_of_;
    const int wave = threadIdx.x >> 6, ln = lane_id();
    const int r0 = blockIdx.x * MS_G;
    if (wave == 0 && ln < MS_G) {
        const int idx = r0 + ln;
        const int rd = perm[idx < n_reads ? idx : r0];
        a_of[ln] = idx < n_reads ? a_plan[rd] : 0;
        n_of[ln] = idx < n_reads ? n_plan[rd] : 0;
        rid_of[ln] = rd;
    }
    __syncthreads();
    const int Sv = ms_ring(cfg.pA_var_window) + 4, Sm = ms_ring(cfg.pA_mean_window) + 4;
    LDS float *base = (LDS float *)ms_raw;
    LDS float *buf_v = base, *out_v = buf_v + MS_G * Sv, *buf_m = out_v + 2 * MS_G * (MS_CHUNK + 4), *out_m = buf_m + MS_G * Sm; // (out: two halves)
    if (wave == 0) ms_chains<true>(sigs, m, a_of, n_of, rid_of, cfg.pA_var_window, series, cap, buf_v, out_v, have);
    else ms_chains<false>(sigs, m, a_of, n_of, rid_of, cfg.pA_mean_window, series, cap, buf_m, out_m, have);
}

static __device__ void row_clear(adp_row *row)
{
    uint32_t *w = reinterpret_cast<uint32_t *>(row);
    for (int i = lane_id(); i < (int)(sizeof(adp_row) / 4); i += 64) w[i] = 0;
    __syncthreads();
    if (lane_id() == 0) { row->n_cand = -1; row->n_open_pores = -1; row->open_pores_more = -1; }
}

static __device__ void row_exception(adp_row *row, int code)
{
    row_clear(row);
    if (lane_id() == 0) { row->success = 0; row->fail_code = code; }
}

// persistent grid: blockIdx.x = slot, block = 64 threads
#ifndef VAL_WPE
#define VAL_WPE 6 // waves per SIMD of k_validate (512 / VAL_WPE vector registers each)
#endif
template <class SIG>
__global__ void __launch_bounds__(64, VAL_WPE) __attribute__((amdgpu_waves_per_eu(VAL_WPE, VAL_WPE))) k_validate(ValidateInT<SIG> in, adp_cfg cfg, adp_row *__restrict__ rows,
                                                 PartReq *__restrict__ preq)
{
    __shared__ __attribute__((aligned(16))) WaveScratch ws_;
    LDS WaveScratch *ws = (LDS WaveScratch *)&ws_;
    LDS SegCache *sc = nullptr; // LDS mirror of the slice: measured slower (occupancy), kept switchable
    const int ln = lane_id();
    float *scr_mean = in.scratch + (size_t)blockIdx.x * 2 * in.scratch_stride;
    float *scr_var = scr_mean + in.scratch_stride;
    for (int r = blockIdx.x; r < in.n_reads; r += gridDim.x) {
        adp_row *row = rows + r;
        row_clear(row);
        if (ln == 0) { preq[r].valid = 0; if (sc) { sc->src = 0; sc->n = 0; } }
        if (in.mbs && in.mbs[r / in.mbsize].status != ADP_MB_OK) continue; // dropped minibatch: zero row
        const typename SIG::Row sig = in.sig.row(r, in.m);
        const long long full_len = in.full_len[r];
        const int S = (int)(full_len < in.m ? full_len : in.m);
        const int64_t *bd = in.bounds + (size_t)r * (1 + in.kmax);
        const long long a_in = bd[0];
        const long long p_in = in.kmax > 0 ? bd[1] : 0;
        const bool topk_none = in.topk_none ? in.topk_none[r] != 0 : false;
        long long a_s = 0, a_e = a_in, p_best = p_in;
        bool p_none = false; // polya_end_best became None (mvs_detect_overwrite)
        int success = 1, fail = ADP_F_NONE, mvs_mask = 0;
        float adapter_med = 0.f, adapter_mad = 0.f;
        bool have_med = false;
        RowW rw{row, 0ull};
        int n_open = -1;

        if (a_e == 0) { success = 0; fail = ADP_F_NO_ADAPTER; }
        else {
            int b = (int)(a_e < S ? a_e : S);
            adapter_med = wave_median(sig, b, 0, 0.f, ws, sc);
            adapter_mad = wave_median(sig, b, 1, adapter_med, ws, sc);
            have_med = true;
        }
        if (success && have_med && adapter_mad != 0.0f &&
            !in_range_d((double)adapter_mad, cfg.adapter_mad_range[0], cfg.adapter_mad_range[1])) {
            success = 0; fail = ADP_F_ADAPTER_MAD;
        }
        if (success && cfg.detect_open_pores && !(g_ablate & 16)) {
            // V2: positions >= 200 pA; keep pos[i] (i >= 1) with a gap >= 10 to pos[i-1]; none kept -> [pos[-1]]
            const int b = (int)(a_e < S ? a_e : S);
            int npos = 0, nvalid = 0, prev_last = -1, lastpos = -1, lastvalid = -1;
            for (int base = 0; base < b; base += 64) {
                int i = base + ln;
                bool f = (i < b) && (200.0f <= sig[i]);
                unsigned long long mk = __ballot(f);
                if (mk) {
                    unsigned long long lower = mk & ((1ull << ln) - 1ull);
                    int prev = lower ? (base + 63 - __clzll((long long)lower)) : prev_last;
                    bool first_overall = (npos == 0) && (lower == 0);
                    bool valid = f && !first_overall && (i - prev >= 10);
                    unsigned long long vm = __ballot(valid);
                    if (valid) {
                        int slot = nvalid + __popcll(vm & ((1ull << ln) - 1ull));
                        if (slot < ADP_MAX_OPEN_PORES) row->open_pores[slot] = i;
                    }
                    if (vm) lastvalid = base + 63 - __clzll((long long)vm);
                    nvalid += __popcll(vm);
                    npos += __popcll(mk);
                    lastpos = base + 63 - __clzll((long long)mk);
                    prev_last = lastpos;
                }
            }
            long long last = -1;
            if (npos == 0) n_open = 0;
            else if (npos == 1 || nvalid == 0) { n_open = 1; last = lastpos; if (ln == 0) row->open_pores[0] = lastpos; }
            else { n_open = nvalid; last = lastvalid; }
            if (n_open > ADP_MAX_OPEN_PORES) {
                // the reference's list has no length limit: the whole of it goes to the call's arena (second scan, rare)
                unsigned int off = 0;
                if (ln == 0) off = atomicAdd(in.op_used, (unsigned int)n_open);
                off = __shfl(off, 0);
                if ((unsigned long long)off + (unsigned long long)n_open <= in.op_cap) {
                    int np2 = 0, nv2 = 0, pl2 = -1;
                    for (int base = 0; base < b; base += 64) {
                        const int i = base + ln;
                        const bool f = (i < b) && (200.0f <= sig[i]);
                        const unsigned long long mk = __ballot(f);
                        if (mk) {
                            const unsigned long long lower = mk & ((1ull << ln) - 1ull);
                            const int prev = lower ? (base + 63 - __clzll((long long)lower)) : pl2;
                            const bool first_overall = (np2 == 0) && (lower == 0);
                            const bool valid = f && !first_overall && (i - prev >= 10);
                            const unsigned long long vm = __ballot(valid);
                            if (valid) in.op_arena[off + nv2 + __popcll(vm & ((1ull << ln) - 1ull))] = i;
                            nv2 += __popcll(vm);
                            np2 += __popcll(mk);
                            pl2 = base + 63 - __clzll((long long)mk);
                        }
                    }
                    if (ln == 0) row->open_pores_more = (int32_t)off;
                } else if (ln == 0) row->open_pores_more = -2; // (the host grows the arena and runs the kernel again)
            }
            if (n_open > 0) {
                a_s = last;
                if (a_e - a_s < cfg.min_obs_adapter) { success = 0; fail = ADP_F_OPEN_PORE; }
            }
        }
        if (success && cfg.real_signal_check && !(g_ablate & 32)) {
            int a = (int)(a_s < S ? a_s : S), b = (int)(a_e < S ? a_e : S);
            int n = b - a; if (n < 0) n = 0;
            const typename SIG::Row x = sig + a;
            bool ok = false;
            if (n >= 2 * cfg.mean_window) {
                float ms = wave_np_mean(x, cfg.mean_window, ws);
                float me = wave_np_mean(x + n - cfg.mean_window, cfg.mean_window, ws);
                rw.set(ADP_C_REAL_MEAN_START, (double)ms);
                rw.set(ADP_C_REAL_MEAN_END, (double)me);
                if (in_range_d((double)ms, cfg.mean_start_range[0], cfg.mean_start_range[1]) &&
                    in_range_d((double)me, cfg.mean_end_range[0], cfg.mean_end_range[1])) {
                    int k = n < cfg.max_obs_local_range ? n : cfg.max_obs_local_range;
                    double lr = VAL_MULTI_RANKS ? wave_local_range(x + n - k, k, ws)
                                                : wave_percentile(x + n - k, k, 85.0, ws, sc) - wave_percentile(x + n - k, k, 15.0, ws, sc);
                    rw.set(ADP_C_REAL_LOCAL_RANGE, lr);
                    ok = in_range_d(lr, cfg.local_range[0], cfg.local_range[1]);
                }
            }
            if (!ok) { success = 0; fail = ADP_F_REAL_RANGE; }
        }
        bool exception = false;
        if (success && cfg.mvs_detect_check && !(g_ablate & 64)) {
            if (p_best == 0) { success = 0; fail = ADP_F_NO_POLYA; }
            else {
                double pr0 = cfg.pA_mean_range[0], pr1 = cfg.pA_mean_range[1];
                int exc = 0;
                if (range_empty(cfg.pA_mean_range) && !range_empty(cfg.pA_mean_adapter_med_scale_range)) {
                    pr0 = cfg.pA_mean_adapter_med_scale_range[0] * (double)adapter_med;
                    pr1 = cfg.pA_mean_adapter_med_scale_range[1] * (double)adapter_med;
                } else if (range_empty(cfg.pA_mean_range)) exc = ADP_F_EXC_PA_RANGE;
                if (!exc && topk_none) exc = ADP_F_EXC_TOPK_NONE;
                if (exc) { row_exception(row, exc); exception = true; }
                long long p_series = 0; // the poly(A) end the precomputed series reach (k_mvs_series)
                if (in.series && in.have_series[r])
                    for (int c = 0; c < in.kmax; c++) { const long long pc = bd[1 + c]; if (pc == 0) break; if (pc > p_series) p_series = pc; }
                // several candidates, no prepared series: run the recurrences ONCE up to the largest candidate (they are
                // causal: every other candidate's series is a prefix) into this slot's scratch
                const float *own_mean = nullptr, *own_var = nullptr;
                float shift_val = 0.f; bool shift_have = false;
                if (p_series == 0 && !exception && !cfg.mvs_detect_overwrite && in.kmax > 1 && bd[1] != 0 && bd[2] != 0 && cfg.pA_var_window <= MV_HIST &&
                    cfg.pA_mean_window <= MV_HIST && cfg.pA_var_window >= 1 && cfg.pA_mean_window >= 1) {
                    long long pmx = 0;
                    for (int c = 0; c < in.kmax; c++) { const long long pc = bd[1 + c]; if (pc == 0) break; if (pc > pmx) pmx = pc; }
                    const int a = (int)(a_e < S ? a_e : S), b = (int)(pmx < S ? pmx : S);
                    if (a_e != 0 && pmx > a_e && b - a >= cfg.pA_var_window && b - a >= cfg.pA_mean_window) {
                        wave_move_series(sig + a, b - a, cfg.pA_var_window, cfg.pA_mean_window, true, true, scr_var, scr_mean, ws);
                        __threadfence_block();
                        __syncthreads();
                        own_mean = scr_mean; own_var = scr_var; p_series = pmx;
                    }
                }
                for (int c = 0; !exception && c < in.kmax; c++) {
                    long long p_e = bd[1 + c];
                    if (p_e == 0) break;
                    if (cfg.mvs_detect_overwrite) {
                        // combined.py:517-562: the adapter end moves to the position the MVS scan finds behind it
                        const MvsLoc o = mvs_detect_at_loc(sig, S, a_e, cfg, pr0, pr1, ws, scr_mean, scr_var, sc, adapter_med);
                        if (o.exc) { row_exception(row, o.exc); exception = true; break; }
                        rw.set(ADP_C_MVS_ADAPTER_END, (double)o.idx);
                        rw.set(ADP_C_MVS_MEAN, o.mean); rw.set(ADP_C_MVS_VAR, o.var);
                        rw.set(ADP_C_MVS_POLYA_MED, o.med); rw.set(ADP_C_MVS_LOCAL_RANGE, o.lrange);
                        rw.set(ADP_C_MVS_MED_SHIFT, o.shift);
                        if (!o.ok) { success = 0; fail = ADP_F_NO_ADAPTER_MVS; break; } // (the other candidates would repeat this evaluation)
                        if (o.idx - a_e > 0) {
                            a_e = o.idx;
                            // Boundaries.polya_end_adjust / .polya_truncated / .trace_early_stop_pos are None on every call
                            // path of v0.2.4: a poly(A) end behind the new adapter end becomes None (combined.py:546-561)
                            if (a_e > p_e) { p_none = true; mvs_mask |= ADP_MVS_TO_EARLY_STOP; }
                        }
                        p_best = p_e;
                        break;
                    }
                    const bool pre = p_series > 0 && p_e <= p_series;
                    const float *pm = !pre ? nullptr : (own_mean ? own_mean : in.series + (size_t)r * 2 * in.series_cap);
                    const float *pv = !pre ? nullptr : (own_var ? own_var : in.series + (size_t)r * 2 * in.series_cap + in.series_cap);
                    MvsOut o = mvs_check(sig, S, a_e, p_e, cfg, pr0, pr1, ws, scr_mean, scr_var, sc, pm, pv,
                                         in.cstat ? in.cstat + (size_t)r * in.kmax + c : nullptr, shift_val, shift_have, pre && !own_mean);
                    if (o.exc) { row_exception(row, o.exc); exception = true; break; }
                    rw.set(ADP_C_MVS_MEAN, o.mean); rw.set(ADP_C_MVS_VAR, o.var);
                    rw.set(ADP_C_MVS_POLYA_MED, o.med); rw.set(ADP_C_MVS_LOCAL_RANGE, o.lrange);
                    rw.set(ADP_C_MVS_MED_SHIFT, o.shift);
                    if (!o.ok) {
                        success = 0; // never reset: later candidates only refresh the reported values
                        if (o.mean == 0) { fail = ADP_F_MVS_NOT_ENOUGH; mvs_mask = 0; }
                        else { fail = ADP_F_MVS_CHECKS; mvs_mask = o.vec_fail; }
                    }
                    if (success) { p_best = p_e; break; }
                }
            }
        }
        if (exception) continue;
        if (success && cfg.detect_med_shift) {
            long long w = cfg.med_shift_window;
            long long r1 = a_e + w; if (r1 > full_len) r1 = full_len; if (r1 > S) r1 = S;
            long long a = a_e < S ? a_e : S;
            long long l0 = a_e - w; if (l0 < 0) l0 = 0; if (l0 > S) l0 = S;
            float sh = wave_median(sig + a, (int)(r1 - a), 0, 0.f, ws, sc) - wave_median(sig + l0, (int)(a - l0), 0, 0.f, ws, sc);
            rw.set(ADP_C_MED_SHIFT, (double)sh);
            if (!in_range_d((double)sh, cfg.med_shift_range[0], cfg.med_shift_range[1])) { success = 0; fail = ADP_F_MED_SHIFT; }
        }
        // S1 partition statistics are computed by k_partition_stats (one 256-thread block per read)
        if (ln == 0) {
            PartReq q;
            q.valid = 1; q.S = S; q.a_s = a_s; q.a_e = a_e; q.p_e = p_best;
            q.adapter_med = adapter_med; q.adapter_mad = adapter_mad;
            q.have_adapter_medmad = (have_med && a_s == 0 && a_e == a_in) ? 1 : 0; q.p_none = p_none ? 1 : 0;
            preq[r] = q;
        }
        rw.set(ADP_C_ADAPTER_END, (double)a_e);
        if (!p_none) rw.set(ADP_C_POLYA_END, (double)p_best);
        rw.set(ADP_C_SIGNAL_LEN, (double)full_len);
        rw.set(ADP_C_PRELOADED, (double)S);
        rw.set(ADP_C_PRIMARY_ADAPTER_END, (double)a_in);
        rw.set(ADP_C_PRIMARY_POLYA_END, (double)p_in);
        if (ln == 0) {
            if (!topk_none) {
                int nc = in.kmax < ADP_MAX_CAND ? in.kmax : ADP_MAX_CAND;
                row->n_cand = nc;
                for (int c = 0; c < nc; c++) row->cand[c] = bd[1 + c];
            }
            row->n_open_pores = n_open;
            row->present = rw.present;
            row->success = success;
            row->fail_code = fail;
            row->mvs_fail_mask = mvs_mask;
        }
    }
}

// ---------------------------------------------------------------- K1 start peak
struct SpOut {
    int32_t valid, flagged_type, has_open_pore, pad;
    int64_t start_peak_idx, next_greater_idx, open_pore_idx;
    float start_peak_pa, next_greater_pa;
};

template <class ROW>
static __device__ __forceinline__ float sp_pooled(ROW row, int m, int ds, int j)
{
    // mean-pool of the RAW signal, zero padded tail, numpy order
    const int b = j * ds;
    return pw_leaf_f32(ds, [&](int k) { int i = b + k; return row.at_or(i, i < m, 0.0f); }) / (float)ds;
}

// 64 pooled values j0 .. j0 + 63 at once: the 64 * ds raw samples are staged in LDS with coalesced 16-byte loads
// (lane-strided scalar loads of ds samples each keep the texture-address unit busy ~4x longer), then lane l pools
// its ds samples from LDS in numpy's order.  Samples at or beyond m count as zero (np.pad).
typedef float sp_f4u __attribute__((ext_vector_type(4), aligned(4)));
template <class ROW>
static __device__ __forceinline__ float sp_pooled_tile(ROW row, int m, int ds, int j0, LDS float *tile)
{
    const int ln = lane_id();
    const long long b0 = (long long)j0 * ds;
    const int nt = 64 * ds;
    ws_sync();
    for (int q = ln * 4; q < nt; q += 256) {
        const long long i = b0 + q;
        float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
        if (i + 3 < m) { const float4 v = row.f4u(i); x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w; }
        else { if (i < m) x0 = row[i]; if (i + 1 < m) x1 = row[i + 1]; if (i + 2 < m) x2 = row[i + 2]; }
        tile[q] = x0; if (q + 1 < nt) tile[q + 1] = x1; if (q + 2 < nt) tile[q + 2] = x2; if (q + 3 < nt) tile[q + 3] = x3;
    }
    ws_sync();
    const LDS float *p = tile + ln * ds;
    return pw_leaf_f32(ds, [&](int k) { return p[k]; }) / (float)ds;
}

// detect_rna_start_peak (reference adapted/detect/start_peak.py:7-119); grid = n_reads waves; dynamic LDS = 64 * ds floats
template <class SIG>
__global__ void __launch_bounds__(64) k_start_peak(SIG sigs, const int32_t *__restrict__ full_len, int n_reads,
                                                   int m, adp_cfg cfg, SpOut *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float sp_tile_raw[];
    LDS float *tile = (LDS float *)sp_tile_raw;
    const int r = blockIdx.x;
    const int ln = lane_id();
    const typename SIG::Row row = sigs.row(r, m);
    const int ds = cfg.sp_downscale_factor;
    const int off1 = cfg.sp_offset1, spmax = cfg.start_peak_max_idx, off2 = cfg.sp_offset2;
    const long long fl = full_len[r];
    const int end_idx = (int)((fl < m ? fl : m) / ds);
    const int L = (m + ds - 1) / ds;
    SpOut o; memset(&o, 0, sizeof(o));
    // open pore: first raw sample above the threshold among the first end_idx RAW samples
    int op = 0x7fffffff;
    const float thr = (float)cfg.open_pore_pa;
    for (int base = 0; base < end_idx && base < m && op == 0x7fffffff; base += 512) { // 8 tiles per round trip to memory
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + u * 64 + ln; v[u] = row.at_or(i, i < end_idx && i < m, -__builtin_inff()); }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned long long mk = __ballot(v[u] > thr);
            if (mk && op == 0x7fffffff) op = base + u * 64 + __ffsll((long long)mk) - 1;
        }
    }
    op = (op == 0x7fffffff) ? 0 : op / ds;
    const bool has_op = op > 0;
    const int a = min(off1, L), b = min(spmax, L);
    bool valid = (b - a > 0);
    float mx = 0.f; int max_idx = 0;
    if (valid) {
        // np.argmax semantics in one sweep: value and FIRST index of the maximum; a NaN anywhere makes the maximum NaN,
        // which equals nothing (index 0)
        float lm = -__builtin_inff(); int li = 0x7fffffff; bool anynan = false;
        for (int j0 = a; j0 < b; j0 += 64) {
            const int j = j0 + ln;
            float v = sp_pooled_tile(row, m, ds, j0, tile);
            if (j >= b) continue;
            if (v != v) anynan = true;
            else if (v > lm || li == 0x7fffffff) { lm = v; li = j; }
        }
        anynan = __any(anynan);
        const float wm = wave_max(lm);
        mx = anynan ? __builtin_nanf("") : wm;
        int first = (!anynan && li != 0x7fffffff && lm == wm) ? li : 0x7fffffff;
        first = wave_min(first);
        max_idx = (first == 0x7fffffff ? 0 : first - a) + off1;
    }
    const int s0 = spmax + off2;
    const int e0 = min(end_idx, L), a2 = min(s0, L);
    if (valid && e0 - a2 <= 0) valid = false;
    int nxt = 0;
    if (valid) {
        int hit = 0x7fffffff;
        for (int base = a2; base < e0 && hit == 0x7fffffff; base += 64) {
            const int j = base + ln;
            const float v = sp_pooled_tile(row, m, ds, base, tile);
            const unsigned long long mk = __ballot(j < e0 && v > mx);
            if (mk) hit = base + __ffsll((long long)mk) - 1;
        }
        nxt = (hit == 0x7fffffff ? 0 : hit - a2) + s0;
        if (nxt >= L) valid = false;
    }
    if (valid) {
        o.valid = 1;
        o.start_peak_idx = (int64_t)max_idx * ds; o.start_peak_pa = mx;
        o.next_greater_idx = (int64_t)nxt * ds; o.next_greater_pa = sp_pooled(row, m, ds, nxt);
        if (has_op) {
            if (fabs((double)nxt - (double)op) <= 2.0 + 0.01 * fabs((double)op)) o.flagged_type = 1;
            else if (max_idx < op && op < nxt) o.flagged_type = 2;
            if (o.flagged_type) { o.has_open_pore = 1; o.open_pore_idx = (int64_t)op * ds; }
        }
    }
    if (ln == 0) out[r] = o;
}

// ---- K1 riding the pooling pass of the LLR path (k_norm_pool<SIG, true>, llr_stream.h) -----------------------------------
// k_sp_head: what lies in front of min_obs_adapter -- the maximum of pooled[offset1 : start_peak_max_idx] and the open-pore scan
// over raw[0 : scan_to) -- and the scan ranges; k_sp_tail: the blocks and samples the pooling pass did not cover (behind
// max_obs_trace, or in front of min_obs_adapter), the flags, the row's SpOut.  Same operations as k_start_peak on the same
// values: the first block above the maximum and the first sample above the open-pore level do not depend on who finds them.
template <class SIG>
__global__ void __launch_bounds__(64) k_sp_head(SIG sigs, const int32_t *__restrict__ full_len, int n_reads, int m, adp_cfg cfg,
                                                int scan_to, SpHead *__restrict__ hd)
{
    extern __shared__ __attribute__((aligned(16))) float sp_tile_raw[];
    LDS float *tile = (LDS float *)sp_tile_raw;
    const int r = blockIdx.x;
    const int ln = lane_id();
    const typename SIG::Row row = sigs.row(r, m);
    const int ds = cfg.sp_downscale_factor;
    const int off1 = cfg.sp_offset1, spmax = cfg.start_peak_max_idx, off2 = cfg.sp_offset2;
    const long long fl = full_len[r];
    const int end_idx = (int)((fl < m ? fl : m) / ds);
    const int L = (m + ds - 1) / ds;
    SpHead o; memset(&o, 0, sizeof(o));
    o.op_end = end_idx < m ? end_idx : m;
    o.op_head = 0x7fffffff; o.op_body = 0x7fffffff; o.hit = 0x7fffffff;
    const float thr = (float)cfg.open_pore_pa;
    const int lim = o.op_end < scan_to ? o.op_end : scan_to;
    int op = 0x7fffffff;
    for (int base = 0; base < lim && op == 0x7fffffff; base += 512) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + u * 64 + ln; v[u] = row.at_or(i, i < lim, -__builtin_inff()); }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned long long mk = __ballot(v[u] > thr);
            if (mk && op == 0x7fffffff) op = base + u * 64 + __ffsll((long long)mk) - 1;
        }
    }
    o.op_head = op;
    const int a = min(off1, L), b = min(spmax, L);
    bool valid = (b - a > 0);
    float mx = 0.f; int max_idx = 0;
    if (valid) {
        float lm = -__builtin_inff(); int li = 0x7fffffff; bool anynan = false;
        for (int j0 = a; j0 < b; j0 += 64) {
            const int j = j0 + ln;
            float v = sp_pooled_tile(row, m, ds, j0, tile);
            if (j >= b) continue;
            if (v != v) anynan = true;
            else if (v > lm || li == 0x7fffffff) { lm = v; li = j; }
        }
        anynan = __any(anynan);
        const float wm = wave_max(lm);
        mx = anynan ? __builtin_nanf("") : wm;
        int first = (!anynan && li != 0x7fffffff && lm == wm) ? li : 0x7fffffff;
        first = wave_min(first);
        max_idx = (first == 0x7fffffff ? 0 : first - a) + off1;
    }
    const int s0 = spmax + off2;
    const int e0 = min(end_idx, L), a2 = min(s0, L);
    if (valid && e0 - a2 <= 0) valid = false;
    o.valid = valid ? 1 : 0; o.mx = mx; o.max_idx = max_idx; o.a2 = a2; o.e0 = e0;
    if (ln == 0) hd[r] = o;
}

// cov0 / cov1: the pooling pass looked at the pooled blocks [cov0, cov1) (inside [a2, e0)) and at the raw samples [cov0 * ds, cov1 * ds)
template <class SIG>
__global__ void __launch_bounds__(64) k_sp_tail(SIG sigs, const int32_t *__restrict__ full_len, int n_reads, int m, adp_cfg cfg,
                                                int cov0, int cov1, const SpHead *__restrict__ hd, SpOut *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float sp_tile_raw[];
    LDS float *tile = (LDS float *)sp_tile_raw;
    const int r = blockIdx.x;
    const int ln = lane_id();
    const typename SIG::Row row = sigs.row(r, m);
    const int ds = cfg.sp_downscale_factor;
    const int spmax = cfg.start_peak_max_idx, off2 = cfg.sp_offset2;
    const int L = (m + ds - 1) / ds;
    const SpHead h = hd[r];
    SpOut o; memset(&o, 0, sizeof(o));
    // open pore: the head's find, else the pooling pass's, else whatever lies behind its range
    int op = h.op_head != 0x7fffffff ? h.op_head : h.op_body;
    const float thr = (float)cfg.open_pore_pa;
    for (int base = cov1 * ds; base < h.op_end && op == 0x7fffffff; base += 512) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + u * 64 + ln; v[u] = row.at_or(i, i < h.op_end, -__builtin_inff()); }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned long long mk = __ballot(v[u] > thr);
            if (mk && op == 0x7fffffff) op = base + u * 64 + __ffsll((long long)mk) - 1;
        }
    }
    op = (op == 0x7fffffff) ? 0 : op / ds;
    const bool has_op = op > 0;
    bool valid = h.valid != 0;
    const float mx = h.mx;
    const int s0 = spmax + off2, a2 = h.a2, e0 = h.e0;
    int nxt = 0;
    if (valid) {
        int hit = 0x7fffffff;
        auto scan = [&](int from, int to) { // first block in [from, to) above the maximum
            for (int base = from; base < to && hit == 0x7fffffff; base += 64) {
                const int j = base + ln;
                const float v = sp_pooled_tile(row, m, ds, base, tile);
                const unsigned long long mk = __ballot(j < to && v > mx);
                if (mk) hit = base + __ffsll((long long)mk) - 1;
            }
        };
        scan(a2, min(e0, max(a2, cov0)));                 // in front of the pooling pass's range
        if (hit == 0x7fffffff) hit = h.hit;              // inside it
        if (hit == 0x7fffffff) scan(max(a2, cov1), e0);  // behind it
        nxt = (hit == 0x7fffffff ? 0 : hit - a2) + s0;
        if (nxt >= L) valid = false;
    }
    if (valid) {
        o.valid = 1;
        o.start_peak_idx = (int64_t)h.max_idx * ds; o.start_peak_pa = mx;
        o.next_greater_idx = (int64_t)nxt * ds; o.next_greater_pa = sp_pooled(row, m, ds, nxt);
        if (has_op) {
            if (fabs((double)nxt - (double)op) <= 2.0 + 0.01 * fabs((double)op)) o.flagged_type = 1;
            else if (h.max_idx < op && op < nxt) o.flagged_type = 2;
            if (o.flagged_type) { o.has_open_pore = 1; o.open_pore_idx = (int64_t)op * ds; }
        }
    }
    if (ln == 0) out[r] = o;
}

// start-peak columns into rows written by k_validate; mode 0: LLR extension (decorate only),
// mode 1: combined_detect_start_peak semantics (flag => failure), any_none => slice TypeError for all
__global__ void k_sp_decorate(const SpOut *__restrict__ sp, adp_row *__restrict__ rows, int n_reads, int mode,
                              const int32_t *__restrict__ any_none)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    adp_row *o = rows + r;
    if (mode == 1 && any_none && any_none[0]) {
        memset(o, 0, sizeof(*o)); o->n_cand = -1; o->n_open_pores = -1; o->open_pores_more = -1; o->success = 0; o->fail_code = ADP_F_EXC_SLICE;
        return;
    }
    if (ADP_F_IS_EXCEPTION(o->fail_code)) return; // (the reference raised: a bare DetectResults)
    SpOut s = sp[r];
    if (!s.valid) return;
    if (mode == 0 && o->present == 0) return; // dropped minibatch
    o->col[ADP_C_SP_IDX] = (double)s.start_peak_idx;
    o->col[ADP_C_SP_PA] = (double)s.start_peak_pa;
    o->col[ADP_C_SP_NEXT_IDX] = (double)s.next_greater_idx;
    o->col[ADP_C_SP_NEXT_PA] = (double)s.next_greater_pa;
    unsigned long long pres = o->present | (1ull << ADP_C_SP_IDX) | (1ull << ADP_C_SP_PA) | (1ull << ADP_C_SP_NEXT_IDX) |
                              (1ull << ADP_C_SP_NEXT_PA);
    if (s.has_open_pore) { o->col[ADP_C_SP_OPEN_PORE_IDX] = (double)s.open_pore_idx; pres |= 1ull << ADP_C_SP_OPEN_PORE_IDX; }
    o->present = pres;
    o->start_peak_type = s.flagged_type;
    if (mode == 1 && s.flagged_type) o->success = 0;
}

// bounds for the start-peak primary: adapter_end = polya_end = next_greater_idx; topk None
__global__ void k_sp_bounds(const SpOut *__restrict__ sp, int n_reads, int64_t *__restrict__ bounds, int8_t *__restrict__ topk_none,
                            int32_t *__restrict__ any_none)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    SpOut s = sp[r];
    bounds[2 * r] = s.valid ? s.next_greater_idx : 0;
    bounds[2 * r + 1] = s.valid ? s.next_greater_idx : 0;
    topk_none[r] = 1;
    if (!s.valid) atomicExch(any_none, 1);
}

// bounds for the LLR primary from pooled-unit indices
// single: combined_detect_llr (combined.py:91-117) stops at an adapter candidate at index 0 -- no poly(A) search
__global__ void k_llr_bounds(const int32_t *__restrict__ adapter_idx, const int32_t *__restrict__ polya_idx, int n_reads, int ds,
                             int off, int64_t *__restrict__ bounds, int8_t *__restrict__ topk_none, int single)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int a = adapter_idx[r], p = polya_idx[r];
    int64_t ae = (a > 0) ? (int64_t)a * ds + off : 0;
    int64_t pe = ((single ? a > 0 : a >= 0) && p > 0) ? (int64_t)p * ds + off : 0;
    bounds[2 * r] = ae;
    bounds[2 * r + 1] = pe;
    topk_none[r] = (pe > 0) ? 0 : 1; // polya_end_topk is only assigned when a poly(A) end was found
}

// ---------------------------------------------------------------- C1 prepare_data (CNN head input)
// reference adapted/detect/cnn.py:70-82: mean-pool the RAW signal from min_obs_adapter on, per-read
// nanmedian / MAD of the pooled values, (x - med) / mad, NaN -> -5.0 (torch.nan_to_num: +-inf ->
// +-FLT_MAX).  Two kernels: k_cnn_pool (a 256-thread block per read; every wave pools tiles of 64 blocks staged in LDS with
// coalesced 16-byte loads, numpy's summation order -- sp_pooled_tile; lane-strided scalar loads ran at 1.8 TB/s) and
// k_cnn_prepare (a wave per read: the two medians and the scaling of the pooled values, in place).
__global__ void __launch_bounds__(256) k_cnn_pool(const float *__restrict__ sigs, int n_reads, int m, int off, int ds, int Lc,
                                                   float *__restrict__ out, int32_t *__restrict__ nan_cnt)
{
    extern __shared__ __attribute__((aligned(16))) float cp_tiles_raw[]; // 4 waves x 64 * ds floats
    const int r = blockIdx.x;
    const int ln = lane_id(), wv = threadIdx.x >> 6;
    LDS float *tile = (LDS float *)cp_tiles_raw + (size_t)wv * 64 * ds;
    const RowF32 row = RowF32{(const GLB float *)sigs + (size_t)r * m + off};
    const int Lseg = m - off;
    float *o = out + (size_t)r * Lc;
    int nn = 0;
    for (int j0 = wv * 64; j0 < Lc; j0 += 256) {
        const float v = sp_pooled_tile(row, Lseg, ds, j0, tile);
        const int j = j0 + ln;
        if (j < Lc) { o[j] = v; if (v != v) nn++; }
    }
    nn = wave_sum(nn);
    if (ln == 0 && nn) atomicAdd(&nan_cnt[r], nn);
}

__global__ void __launch_bounds__(64) k_cnn_prepare(int n_reads, int Lc, float *__restrict__ out, const int32_t *__restrict__ nan_cnt)
{
    __shared__ __attribute__((aligned(16))) WaveScratch ws_;
    const int r = blockIdx.x;
    const int ln = lane_id();
    float *o = out + (size_t)r * Lc;
    const int n = Lc - nan_cnt[r];
    LDS WaveScratch *ws = (LDS WaveScratch *)&ws_;
    // NaNs after the last valid block (a short read) leave o[0..n) NaN-free; NaNs inside the read (NaN samples in
    // the signal) are stepped over by the selection instead: np.nanmedian either way
    bool tail = true;
    for (int j = n + ln; j < Lc; j += 64) tail &= o[j] != o[j];
    tail = __all(tail) || n <= 0;
    float med, mad;
    if (tail) {
        med = wave_median(o, n, 0, 0.f, ws);
        mad = wave_median(o, n, 1, med, ws);
    } else {
        float vk, vkm1;
        wave_select2_skipnan(o, Lc, n / 2, 0, 0.f, ws, vk, vkm1);
        med = (n & 1) ? vk : (vkm1 + vk) / 2.0f;
        wave_select2_skipnan(o, Lc, n / 2, 1, med, ws, vk, vkm1);
        mad = (n & 1) ? vk : (vkm1 + vk) / 2.0f;
    }
    __syncthreads();
    for (int j = ln; j < Lc; j += 64) {
        float v = (o[j] - med) / mad;
        if (v != v) v = -5.0f;
        else if (__builtin_isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
        o[j] = v;
    }
}
