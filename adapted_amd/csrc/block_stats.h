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
};

static __device__ __forceinline__ float bs_x2(float x, int mode, float c)
{
    if (mode == 0) return x;
    float d = x - c;
    return d * d;
}

// numpy-ordered sum of xf(x[0..n)) with a per-element side effect `side(raw x)`; all threads return the sum
template <class Side>
static __device__ float block_np_sum(const float *__restrict__ x, int n, int mode, float c, BlockScratch *bs, Side side)
{
    const int tid = threadIdx.x;
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
                side(v[u]);
                bs->u.stage[(e >> 7) * BS_LEAF_STRIDE + (e & 127)] = bs_x2(v[u], mode, c);
            }
            __syncthreads();
            const float *q = bs->u.stage + (tid >> 3) * BS_LEAF_STRIDE + (tid & 7);
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
    if (tail > 0) {
        for (int i = tid; i < tail; i += BS_THREADS) side(x[s + i]);
        __syncthreads();
        if (tid < 64) {
            int id = 0;
            ws_enum_leaves(x + s, 0, tail, mode, c, &bs->u.ws, id);
        }
        __syncthreads();
        if (tid < 64) {
            int id2 = 0;
            total += ws_eval_tree(tail, &bs->u.ws, id2);
        }
    }
    __syncthreads();
    if (tid == 0) bs->bcast[0] = total;
    __syncthreads();
    float res = bs->bcast[0];
    __syncthreads();
    return res;
}

static __device__ __forceinline__ uint32_t bs_window_lo(float pivot)
{
    uint32_t k18 = f2key(pivot) >> 14;
    uint32_t oct = k18 & ~511u;
    return oct >= 1024u ? oct - 1024u : 0u;
}

// locate the bucket holding rank k in hist (counts below the window in `under`); sets bs->bin/before,
// bs->flag = 1 if the rank lies outside the window
static __device__ void block_find_bin(BlockScratch *bs, int k, int under)
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
static __device__ __noinline__ float bs_median_from_bucket(BlockScratch *bs, int n, int rk, uint32_t below_key)
{
    // called by all threads; wave 0 computes, result broadcast
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid < 64) {
        // the staging area doubles as the wave scratch; the bucket lives in bs->collect (LDS, generic pointer)
        float vk, vkm1;
        wave_select2(bs->collect, bs->ncollect, rk, 0, 0.f, &bs->u.ws, vk, vkm1);
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
static __device__ SegStats block_segment_stats(const float *__restrict__ x, int n, BlockScratch *bs, bool have_medmad,
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
    int under = 0;
    float sum = block_np_sum(x, n, 0, 0.f, bs, [&](float v) {
        if (have_medmad) return;
        uint32_t k18 = f2key(v) >> 14;
        if (k18 < wlo) under++;
        else if (k18 - wlo < (uint32_t)BS_BINS) atomicAdd(&bs->hist[k18 - wlo], 1u);
    });
    o.mean = sum / (float)n;
    bool fallback_med = false;
    int bin = 0, rk = 0;
    if (!have_medmad) {
        // block-wide count of samples below the window
        under = wave_sum(under);
        if ((tid & 63) == 0) bs->scan[4 + (tid >> 6)] = under;
        __syncthreads();
        under = bs->scan[4] + bs->scan[5] + bs->scan[6] + bs->scan[7];
        block_find_bin(bs, k1, under);
        fallback_med = bs->flag != 0;
        bin = bs->bin; rk = k1 - bs->before;
        __syncthreads();
        if (tid == 0) { bs->ncollect = 0; bs->below = 0; }
        __syncthreads();
    }
    // ---- pass 2: variance + collect the median bucket ----------------------------------------
    uint32_t below = 0;
    const uint32_t target = wlo + (uint32_t)bin;
    float sum2 = block_np_sum(x, n, 2, o.mean, bs, [&](float v) {
        if (have_medmad || fallback_med) return;
        uint32_t key = f2key(v);
        uint32_t k18 = key >> 14;
        if (k18 == target) { int slot = atomicAdd(&bs->ncollect, 1); if (slot < BS_COLLECT) bs->collect[slot] = v; }
        else if (k18 < target && key > below) below = key;
    });
    o.sd = sqrtf(sum2 / (float)n);
    if (have_medmad) { o.med = med_in; o.mad = mad_in; return o; }
    below = wave_max(below);
    if ((tid & 63) == 0 && below) atomicMax(&bs->below, below);
    __syncthreads();
    if (fallback_med || bs->ncollect > BS_COLLECT) {
        __syncthreads();
        if (tid < 64) { float m_ = wave_median(x, n, 0, 0.f, &bs->u.ws); if (tid == 0) bs->bcast[1] = m_; }
        __syncthreads();
        o.med = bs->bcast[1];
        __syncthreads();
    } else {
        o.med = bs_median_from_bucket(bs, n, rk, bs->below);
    }
    // ---- pass 3: histogram of |x - med| around 0.6745 * sd ------------------------------------
    const float med = o.med;
    float pivot = 0.6745f * o.sd;
    if (!(pivot > 0.f)) pivot = 1.0f;
    wlo = bs_window_lo(pivot);
    for (int i = tid; i < BS_BINS; i += BS_THREADS) bs->hist[i] = 0;
    __syncthreads();
    under = 0;
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
                else if (k18 - wlo < (uint32_t)BS_BINS) atomicAdd(&bs->hist[k18 - wlo], 1u);
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
        below = 0;
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
                    if (k18 == tgt) { int slot = atomicAdd(&bs->ncollect, 1); if (slot < BS_COLLECT) bs->collect[slot] = d; }
                    else if (k18 < tgt && key > below) below = key;
                }
            }
        }
        below = wave_max(below);
        if ((tid & 63) == 0 && below) atomicMax(&bs->below, below);
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
    __shared__ BlockScratch bs;
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
        if (n <= 0) s.mean = s.sd = s.med = s.mad = __builtin_nanf("");
        else s = block_segment_stats(sig + a, n, &bs, p == 0 && q.have_adapter_medmad, q.adapter_med, q.adapter_mad);
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
