// block_stats.h -- S1 calc_partition_stats (mean, std, median, MAD of a read segment) by one
// 256-thread workgroup per read, in FOUR passes over the segment instead of ten:
//
//   pass 1  numpy-ordered float32 sum (-> mean)   +  18-bit key histogram around a pivot (median bucket)
//   pass 2  numpy-ordered sum of (x-mean)^2 (-> var)  +  collect the median bucket + max key below it
//   pass 3  18-bit histogram of |x - med|
//   pass 4  collect the MAD bucket + max key below it
//
// reference: adapted/partition/signal_partitions.py:81-96 (np.mean, np.std, np.median, np.median(|x-med|)
// on float32 slices).  Sums follow numpy's add.reduce association exactly (8192-element chunks in
// sequence; each chunk a balanced tree over 128-element leaves; each leaf 8 interleaved accumulators):
// a half chunk (4096 samples) is staged in LDS with coalesced loads, 256 threads each run one
// accumulator chain of 16 samples, xor-shuffles fold the 8 accumulators and then the 64 leaves.
// Selection is exact: the histogram is over the top 18 bits of the order-preserving key inside a
// 4-octave window around a pivot (0.125 pA bins near 100 pA); the bucket holding the wanted rank is
// copied to LDS (<= 2048 samples) and finished there.  A rank outside the window or an overflowing
// bucket falls back to the generic 4-pass radix select (wave_stats.h) -- slower, never wrong.
#pragma once
#include "common.h"
#include "wave_stats.h"

#define BS_THREADS 256
#define BS_LEAF_STRIDE 136
#define BS_BINS 2048
#define BS_COLLECT 2048

// (defined in adapted_hip.hip) timing experiments only (ADP_ABLATE); results are wrong when non-zero
extern __device__ int g_ablate;

struct BlockScratch {
    union {
        float stage[32 * BS_LEAF_STRIDE];
        WaveScratch ws; // generic wave-level fallbacks reuse the staging area
    } u;
    uint32_t hist[BS_BINS];
    float collect[BS_COLLECT];
    float leafsum[64];
    int scan[8];
    int bin, before, ncollect, flag;
    uint32_t below;
    float bcast[4];
    int nleaf;
    short leaf_off[132], leaf_len[132]; // numpy's pairwise leaves of a ragged (< 8192) chunk
};

static __device__ __forceinline__ float bs_x2(float x, int mode, float c)
{
    if (mode == 0) return x;
    float d = x - c;
    return d * d;
}

// numpy's pairwise recursion over a ragged chunk (< 8192 samples, depth <= 7), unrolled at compile time so
// that no private stack is needed: leaves left to right, then the same tree over the leaf sums.
template <int D>
static __device__ __forceinline__ void bs_enum_leaves(int off, int len, LDS BlockScratch *bs, int &nl)
{
    if (D == 0 || len <= 128) { bs->leaf_off[nl] = (short)off; bs->leaf_len[nl] = (short)len; nl++; return; }
    int n2 = len / 2;
    n2 -= n2 % 8;
    bs_enum_leaves<(D > 0 ? D - 1 : 0)>(off, n2, bs, nl);
    bs_enum_leaves<(D > 0 ? D - 1 : 0)>(off + n2, len - n2, bs, nl);
}
template <int D>
static __device__ __forceinline__ float bs_eval_tree(int len, const LDS float *leaf, int &id)
{
    if (D == 0 || len <= 128) return leaf[id++];
    int n2 = len / 2;
    n2 -= n2 % 8;
    float a = bs_eval_tree<(D > 0 ? D - 1 : 0)>(n2, leaf, id);
    float b = bs_eval_tree<(D > 0 ? D - 1 : 0)>(len - n2, leaf, id);
    return a + b;
}

static __device__ __noinline__ void bs_enum_tail(int tail, LDS BlockScratch *bs)
{
    int nl = 0;
    bs_enum_leaves<7>(0, tail, bs, nl);
    bs->nleaf = nl;
}
static __device__ __noinline__ float bs_eval_tail(int tail, const LDS float *leaf)
{
    int id = 0;
    return bs_eval_tree<7>(tail, leaf, id);
}

enum { SIDE_NONE = 0, SIDE_HIST = 1, SIDE_COLLECT = 2 };
struct SumAux { float sum; uint32_t aux; };

// per-sample side effect of a summing pass (kept in registers: no captured state)
//   SIDE_HIST:    18-bit key histogram inside the window starting at `param`; aux counts samples below it
//   SIDE_COLLECT: copy the samples of bucket `param` to LDS; aux tracks the largest key below the bucket
template <int SIDE>
static __device__ __forceinline__ void bs_side(float v, uint32_t param, LDS BlockScratch *bs, uint32_t &aux)
{
    if (SIDE == SIDE_HIST) {
        uint32_t k18 = f2key(v) >> 14;
        if (k18 < param) aux++;
        else if (k18 - param < (uint32_t)BS_BINS)
            __hip_atomic_fetch_add(&bs->hist[k18 - param], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (SIDE == SIDE_COLLECT) {
        uint32_t key = f2key(v);
        uint32_t k18 = key >> 14;
        if (k18 == param) {
            int slot = __hip_atomic_fetch_add(&bs->ncollect, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (slot < BS_COLLECT) bs->collect[slot] = v;
        } else if (k18 < param && key > aux) aux = key;
    }
}

// numpy-ordered sum of xf(x[0..n)) fused with a per-sample side effect; all threads return the sum and the
// block-reduced aux (SIDE_HIST: count below the window; SIDE_COLLECT: max key below the bucket)
template <int SIDE>
static __device__ __noinline__ SumAux block_np_sum(const float *__restrict__ x, int n, int mode, float c, LDS BlockScratch *bs,
                                                   uint32_t param)
{
    const int tid = threadIdx.x;
    uint32_t aux = 0;
    float total = 0.0f; // meaningful in wave 0
    int s = 0;
    for (; s + 8192 <= n; s += 8192) {
        for (int half = 0; half < 2; half++) {
            const float *p = x + s + half * 4096;
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = p[u * 256 + tid];
            __syncthreads(); // previous chain reads of the staging area are done
#pragma unroll
            for (int u = 0; u < 16; u++) {
                int e = u * 256 + tid;
                bs_side<SIDE>(v[u], param, bs, aux);
                bs->u.stage[(e >> 7) * BS_LEAF_STRIDE + (e & 127)] = bs_x2(v[u], mode, c);
            }
            __syncthreads();
            const LDS float *q = bs->u.stage + (tid >> 3) * BS_LEAF_STRIDE + (tid & 7);
            float r = q[0];
#pragma unroll
            for (int t = 1; t < 16; t++) r += q[8 * t];
            r = r + __shfl_xor(r, 1);
            r = r + __shfl_xor(r, 2);
            r = r + __shfl_xor(r, 4);
            if ((tid & 7) == 0) bs->leafsum[half * 32 + (tid >> 3)] = r;
        }
        __syncthreads();
        if (tid < 64) {
            float l = bs->leafsum[tid];
            for (int o = 1; o < 64; o <<= 1) l = l + __shfl_xor(l, o);
            total += l;
        }
    }
    const int tail = n - s;
    if (tail > 0 && !(g_ablate & 4)) {
        for (int i = tid; i < tail; i += BS_THREADS) bs_side<SIDE>(x[s + i], param, bs, aux);
        // leaves of numpy's pairwise recursion over the ragged chunk (split n -> n2 = (n/2) & ~7, n - n2)
        if (tid == 0) bs_enum_tail(tail, bs);
        __syncthreads();
        const int nleaf = bs->nleaf;
        const float *xt = x + s;
        for (int g0 = 0; g0 < nleaf; g0 += 32) {
            // stage 32 leaves: wave w loads leaves g0 + 8w .. g0 + 8w + 7 (coalesced, 2 loads per leaf)
            __syncthreads();
            const int w = tid >> 6, ln = tid & 63;
            for (int q = 0; q < 8; q++) {
                int l = g0 + w * 8 + q;
                if (l < nleaf) {
                    int off = bs->leaf_off[l], len = bs->leaf_len[l];
                    LDS float *dst = bs->u.stage + (w * 8 + q) * BS_LEAF_STRIDE;
                    if (ln < len) dst[ln] = bs_x2(xt[off + ln], mode, c);
                    if (ln + 64 < len) dst[ln + 64] = bs_x2(xt[off + ln + 64], mode, c);
                }
            }
            __syncthreads();
            // thread (leaf = tid >> 3, j = tid & 7): accumulator chain j of numpy's 8-accumulator leaf
            const int ll = tid >> 3, j = tid & 7;
            const int l = g0 + ll;
            const int len = (l < nleaf) ? bs->leaf_len[l] : 0;
            const LDS float *q = bs->u.stage + ll * BS_LEAF_STRIDE;
            float r = 0.0f;
            if (len >= 8) {
                r = q[j];
                const int lim = len - (len % 8);
                for (int i = 8; i < lim; i += 8) r += q[i + j];
            }
            r = r + __shfl_xor(r, 1);
            r = r + __shfl_xor(r, 2);
            r = r + __shfl_xor(r, 4);
            if (j == 0 && l < nleaf) {
                float res;
                if (len >= 8) { res = r; for (int i = len - (len % 8); i < len; i++) res += q[i]; }
                else { res = 0.0f; for (int i = 0; i < len; i++) res += q[i]; }
                bs->u.ws.leaf[l] = res;
            }
        }
        __syncthreads();
        if (tid < 64) total += bs_eval_tail(tail, bs->u.ws.leaf);
    }
    __syncthreads();
    if (tid == 0) { bs->bcast[0] = total; bs->below = 0; }
    __syncthreads();
    if (SIDE == SIDE_HIST) {
        uint32_t w = (uint32_t)wave_sum((int)aux);
        if ((tid & 63) == 0 && w) __hip_atomic_fetch_add(&bs->below, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (SIDE == SIDE_COLLECT) {
        uint32_t w = wave_max(aux);
        if ((tid & 63) == 0 && w) __hip_atomic_fetch_max(&bs->below, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    SumAux r;
    r.sum = bs->bcast[0];
    r.aux = bs->below;
    __syncthreads();
    return r;
}

static __device__ __forceinline__ uint32_t bs_window_lo(float pivot)
{
    uint32_t k18 = f2key(pivot) >> 14;
    uint32_t oct = k18 & ~511u;
    return oct >= 1024u ? oct - 1024u : 0u;
}

// locate the bucket holding rank k in hist (counts below the window in `under`); sets bs->bin/before,
// bs->flag = 1 if the rank lies outside the window
static __device__ __noinline__ void block_find_bin(LDS BlockScratch *bs, int k, int under)
{
    const int tid = threadIdx.x;
    uint32_t h[8];
    int s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { h[j] = bs->hist[tid * 8 + j]; s += (int)h[j]; }
    int incl = wave_scan_incl(s);
    if ((tid & 63) == 63) bs->scan[tid >> 6] = incl;
    if (tid == 0) { bs->flag = 1; bs->bin = 0; bs->before = 0; }
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (tid >> 6); w++) woff += bs->scan[w];
    int excl = under + woff + incl - s;
    if (k >= excl && k < excl + s) {
        int cacc = excl;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (k >= cacc && k < cacc + (int)h[j]) { bs->bin = tid * 8 + j; bs->before = cacc; bs->flag = 0; }
            cacc += (int)h[j];
        }
    }
    __syncthreads();
}

struct SegStats { float mean, sd, med, mad; };

// exact k-th / (k-1)-th from a collected bucket (wave 0), given the max key below the bucket
static __device__ __noinline__ float bs_median_from_bucket(LDS BlockScratch *bs, int n, int rk, uint32_t below_key)
{
    // called by all threads; wave 0 computes, result broadcast
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid < 64) {
        // the staging area doubles as the wave scratch; the bucket lives in bs->collect (LDS, generic pointer)
        float vk, vkm1;
        wave_select2_lds(bs->collect, bs->ncollect, rk, 0, 0.f, &bs->u.ws, vk, vkm1);
        float res = vk;
        if ((n & 1) == 0) {
            float lo = (rk >= 1) ? vkm1 : key2f(below_key);
            res = (lo + vk) / 2.0f;
        }
        if (tid == 0) bs->bcast[1] = res;
    }
    __syncthreads();
    float r = bs->bcast[1];
    __syncthreads();
    return r;
}

// mean / std / median / MAD of x[0..n), n >= 1.  have_medmad: reuse med_in / mad_in (adapter partition).
static __device__ SegStats block_segment_stats(const float *__restrict__ x, int n, LDS BlockScratch *bs, bool have_medmad,
                                               float med_in, float mad_in)
{
    const int tid = threadIdx.x;
    SegStats o;
    const int k1 = n / 2;
    // ---- pass 1: mean + median histogram ------------------------------------------------
    uint32_t wlo = 0;
    if (!have_medmad) {
        float a = x[n / 4], b = x[n / 2], c3 = x[(3 * (long long)n) / 4];
        float pivot = fmaxf(fminf(a, b), fminf(fmaxf(a, b), c3));
        wlo = bs_window_lo(pivot);
    }
    for (int i = tid; i < BS_BINS; i += BS_THREADS) bs->hist[i] = 0;
    __syncthreads();
    SumAux p1 = have_medmad ? block_np_sum<SIDE_NONE>(x, n, 0, 0.f, bs, 0u) : block_np_sum<SIDE_HIST>(x, n, 0, 0.f, bs, wlo);
    o.mean = p1.sum / (float)n;
    bool fallback_med = false;
    int bin = 0, rk = 0;
    if (!have_medmad) {
        block_find_bin(bs, k1, (int)p1.aux);
        fallback_med = bs->flag != 0;
        bin = bs->bin; rk = k1 - bs->before;
        __syncthreads();
        if (tid == 0) bs->ncollect = 0;
        __syncthreads();
    }
    // ---- pass 2: variance + collect the median bucket ----------------------------------------
    const uint32_t target = wlo + (uint32_t)bin;
    SumAux p2 = (have_medmad || fallback_med) ? block_np_sum<SIDE_NONE>(x, n, 2, o.mean, bs, 0u)
                                              : block_np_sum<SIDE_COLLECT>(x, n, 2, o.mean, bs, target);
    o.sd = sqrtf(p2.sum / (float)n);
    if (have_medmad) { o.med = med_in; o.mad = mad_in; return o; }
    if (fallback_med || bs->ncollect > BS_COLLECT) {
        __syncthreads();
        if (tid < 64) { float m_ = wave_median(x, n, 0, 0.f, &bs->u.ws); if (tid == 0) bs->bcast[1] = m_; }
        __syncthreads();
        o.med = bs->bcast[1];
        __syncthreads();
    } else {
        o.med = bs_median_from_bucket(bs, n, rk, p2.aux);
    }
    if (g_ablate & 2) { o.mad = 0; return o; }
    // ---- pass 3: histogram of |x - med| around 0.6745 * sd ------------------------------------
    const float med = o.med;
    float pivot = 0.6745f * o.sd;
    if (!(pivot > 0.f)) pivot = 1.0f;
    wlo = bs_window_lo(pivot);
    for (int i = tid; i < BS_BINS; i += BS_THREADS) bs->hist[i] = 0;
    __syncthreads();
    int under = 0;
    for (int base = 0; base < n; base += BS_THREADS * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { int i = base + u * BS_THREADS + tid; v[u] = (i < n) ? x[i] : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int i = base + u * BS_THREADS + tid;
            if (i < n) {
                uint32_t k18 = f2key(fabsf(v[u] - med)) >> 14;
                if (k18 < wlo) under++;
                else if (k18 - wlo < (uint32_t)BS_BINS) __hip_atomic_fetch_add(&bs->hist[k18 - wlo], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    under = wave_sum(under);
    __syncthreads();
    if ((tid & 63) == 0) bs->scan[4 + (tid >> 6)] = under;
    __syncthreads();
    under = bs->scan[4] + bs->scan[5] + bs->scan[6] + bs->scan[7];
    block_find_bin(bs, k1, under);
    const bool fallback_mad = bs->flag != 0;
    bin = bs->bin; rk = k1 - bs->before;
    __syncthreads();
    if (tid == 0) { bs->ncollect = 0; bs->below = 0; }
    __syncthreads();
    // ---- pass 4: collect the MAD bucket ----------------------------------------------------------
    if (!fallback_mad) {
        const uint32_t tgt = wlo + (uint32_t)bin;
        uint32_t below = 0;
        for (int base = 0; base < n; base += BS_THREADS * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { int i = base + u * BS_THREADS + tid; v[u] = (i < n) ? x[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int i = base + u * BS_THREADS + tid;
                if (i < n) {
                    float d = fabsf(v[u] - med);
                    uint32_t key = f2key(d);
                    uint32_t k18 = key >> 14;
                    if (k18 == tgt) { int slot = __hip_atomic_fetch_add(&bs->ncollect, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (slot < BS_COLLECT) bs->collect[slot] = d; }
                    else if (k18 < tgt && key > below) below = key;
                }
            }
        }
        below = wave_max(below);
        if ((tid & 63) == 0 && below) __hip_atomic_fetch_max(&bs->below, below, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
    }
    if (fallback_mad || bs->ncollect > BS_COLLECT) {
        __syncthreads();
        if (tid < 64) { float m_ = wave_median(x, n, 1, med, &bs->u.ws); if (tid == 0) bs->bcast[1] = m_; }
        __syncthreads();
        o.mad = bs->bcast[1];
        __syncthreads();
    } else {
        o.mad = bs_median_from_bucket(bs, n, rk, bs->below);
    }
    return o;
}

// per-read partition request left behind by k_validate
struct PartReq {
    int32_t valid;      // 0: skip (dropped minibatch or exception row)
    int32_t S;          // samples available: min(full_len, m)
    int64_t a_s, a_e, p_e;
    float adapter_med, adapter_mad;
    int32_t have_adapter_medmad, pad;
};

// grid = n_reads blocks of 256 threads
__global__ void __launch_bounds__(BS_THREADS, 4) k_partition_stats(const float *__restrict__ sigs, int m, const PartReq *__restrict__ req,
                                                               adp_row *__restrict__ rows)
{
    __shared__ BlockScratch bs_;
    LDS BlockScratch *bs = (LDS BlockScratch *)&bs_;
    const int r = blockIdx.x;
    const PartReq q = req[r];
    if (!q.valid) return;
    const float *sig = sigs + (size_t)r * m;
    adp_row *row = rows + r;
    const int S = q.S;
    unsigned long long present = 0;
    const long long starts[3] = {q.a_s, q.a_e, q.p_e};
    const long long ends[3] = {q.a_e, q.p_e, (long long)S};
    const int c_start[3] = {ADP_C_ADAPTER_START, ADP_C_POLYA_START, ADP_C_RNA_START};
    const int c_len[3] = {ADP_C_ADAPTER_LEN, ADP_C_POLYA_LEN, ADP_C_RNA_LEN};
    for (int p = 0; p < 3; p++) {
        const long long st = starts[p], en = ends[p];
        if (threadIdx.x == 0) row->col[c_start[p]] = (double)st;
        present |= 1ull << c_start[p];
        if (en <= st) continue;
        long long a = st < S ? st : S, b = en < S ? en : S;
        int n = (int)(b - a);
        SegStats s;
        if (n <= 0 || ((g_ablate & 8) && p < 2)) s.mean = s.sd = s.med = s.mad = __builtin_nanf("");
        else s = block_segment_stats(sig + a, n, bs, p == 0 && q.have_adapter_medmad, q.adapter_med, q.adapter_mad);
        if (threadIdx.x == 0) {
            row->col[c_len[p]] = (double)(en - st);
            row->col[c_len[p] + 1] = (double)s.mean;
            row->col[c_len[p] + 2] = (double)s.sd;
            row->col[c_len[p] + 3] = (double)s.med;
            row->col[c_len[p] + 4] = (double)s.mad;
        }
        present |= 31ull << c_len[p];
        __syncthreads();
    }
    if (threadIdx.x == 0) row->present |= present;
}
