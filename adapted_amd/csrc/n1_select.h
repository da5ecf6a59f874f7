// n1_select.h -- N1: exact batch-global nanmedian / MAD of batch[:, :T] per minibatch.
//
// Replaces normalize_signal/med_mad (reference adapted/detect/normalize.py:15-63) as called
// with the whole 2-D minibatch at adapted/detect/combined.py:128-132: ONE median and ONE MAD
// per minibatch, computed exactly (k-th order statistics; even count -> float32 mean of the
// two middle values, like numpy).
//
// Method: MSB-first radix select on order-preserving uint32 keys, three HBM passes of
// 11/11/10 bits.  Each pass builds a per-block histogram in LDS (float4 loads, per-lane
// aggregation of equal digits to tame same-address LDS atomics on the clustered top digit)
// and flushes it to one global histogram per minibatch.  The lower median of an even count
// costs no extra pass: pass p also tracks the maximum key that lies below the bucket chosen
// in pass p-1 (one atomicMax per block).
#pragma once
#include "common.h"

#define N1_BINS 2048
#define N1_THREADS 256

// transform: mode 0 -> x ; mode 1 -> |x - med| (float32, as numpy computes np.abs(signal - med))
static __device__ __forceinline__ float n1_xform(float x, int mode, float med) { return mode ? fabsf(x - med) : x; }

template <int PASS>
static __device__ __forceinline__ void n1_account(float x, int mode, float med, uint32_t prefix, uint32_t *hist,
                                                   uint32_t &below, uint32_t &d_prev, uint32_t &run)
{
    float v = n1_xform(x, mode, med);
    if (v != v) return;
    uint32_t key = f2key(v);
    uint32_t digit;
    if (PASS == 0) digit = key >> 21;
    else if (PASS == 1) {
        uint32_t top = key >> 21;
        if (top != prefix) { if (top < prefix && key > below) below = key; return; }
        digit = (key >> 10) & 2047u;
    } else {
        uint32_t top = key >> 10;
        if (top != prefix) { if (top < prefix && key > below) below = key; return; }
        digit = key & 1023u;
    }
    // run-length aggregation inside the lane
    if (digit == d_prev) run++;
    else { if (run) atomicAdd(&hist[d_prev], run); d_prev = digit; run = 1; }
}

// grid = (blocks_per_minibatch, n_minibatch); block = N1_THREADS
template <int PASS>
__global__ void __launch_bounds__(N1_THREADS) k_n1_hist(const float *__restrict__ sig, int n_reads, int m, int T, int mbsize,
                                                         int mode, const MbState *__restrict__ mbs, uint32_t *__restrict__ ghist,
                                                         uint32_t *__restrict__ gbelow)
{
    __shared__ uint32_t hist[N1_BINS];
    __shared__ uint32_t sbelow;
    const int mb = blockIdx.y;
    const MbState st = mbs[mb];
    if (st.status != ADP_MB_OK) return;
    for (int i = threadIdx.x; i < N1_BINS; i += N1_THREADS) hist[i] = 0;
    if (threadIdx.x == 0) sbelow = 0;
    __syncthreads();
    const int r0 = mb * mbsize;
    const int r1 = min(n_reads, r0 + mbsize);
    const float med = st.med;
    const uint32_t prefix = st.prefix;
    uint32_t below = 0, d_prev = 0xffffffffu, run = 0;
    const bool vec = ((m & 3) == 0) && ((reinterpret_cast<uintptr_t>(sig) & 15) == 0);
    for (int r = r0 + blockIdx.x; r < r1; r += gridDim.x) {
        const float *row = sig + (size_t)r * m;
        if (vec) {
            const int T4 = T >> 2;
            const float4 *row4 = reinterpret_cast<const float4 *>(row);
            for (int i = threadIdx.x; i < T4; i += N1_THREADS) {
                float4 v = row4[i];
                n1_account<PASS>(v.x, mode, med, prefix, hist, below, d_prev, run);
                n1_account<PASS>(v.y, mode, med, prefix, hist, below, d_prev, run);
                n1_account<PASS>(v.z, mode, med, prefix, hist, below, d_prev, run);
                n1_account<PASS>(v.w, mode, med, prefix, hist, below, d_prev, run);
            }
            for (int i = (T4 << 2) + threadIdx.x; i < T; i += N1_THREADS)
                n1_account<PASS>(row[i], mode, med, prefix, hist, below, d_prev, run);
        } else {
            for (int i = threadIdx.x; i < T; i += N1_THREADS)
                n1_account<PASS>(row[i], mode, med, prefix, hist, below, d_prev, run);
        }
    }
    if (run) atomicAdd(&hist[d_prev], run);
    if (PASS > 0) {
        below = wave_max(below);
        if (lane_id() == 0 && below) atomicMax(&sbelow, below);
    }
    __syncthreads();
    uint32_t *gh = ghist + (size_t)mb * N1_BINS;
    for (int i = threadIdx.x; i < N1_BINS; i += N1_THREADS) {
        uint32_t c = hist[i];
        if (c) atomicAdd(&gh[i], c);
    }
    if (PASS > 0 && threadIdx.x == 0 && sbelow) atomicMax(&gbelow[mb], sbelow);
}

// One block per minibatch: locate the bucket of the running rank, update the state, clear the
// histogram for the next pass.  PASS 2 finishes the selection and writes med (mode 0) or
// mad + clip bounds (mode 1).
template <int PASS>
__global__ void __launch_bounds__(256) k_n1_pick(MbState *__restrict__ mbs, uint32_t *__restrict__ ghist,
                                                 uint32_t *__restrict__ gbelow, int mode, double thresh)
{
    __shared__ unsigned long long part[256];
    __shared__ int s_bin;
    __shared__ unsigned long long s_before;
    const int mb = blockIdx.x;
    MbState st = mbs[mb];
    if (st.status != ADP_MB_OK) return;
    uint32_t *gh = ghist + (size_t)mb * N1_BINS;
    const int nb = (PASS == 2) ? 1024 : 2048;
    const int per = nb / 256;
    uint32_t loc[8];
    unsigned long long s = 0;
    for (int j = 0; j < per; j++) { loc[j] = gh[threadIdx.x * per + j]; s += loc[j]; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 256; i++) { unsigned long long t = part[i]; part[i] = tot; tot += t; }
        if (PASS == 0) {
            st.n_valid = tot;
            st.krem = tot / 2; // rank of the upper median
        }
        s_before = 0;
        s_bin = -1;
        mbs[mb].n_valid = st.n_valid;
    }
    __syncthreads();
    if (PASS == 0) st.n_valid = mbs[mb].n_valid, st.krem = st.n_valid / 2;
    if (st.n_valid == 0) {
        if (threadIdx.x == 0) {
            mbs[mb].med = __builtin_nanf(""); mbs[mb].mad = __builtin_nanf("");
            mbs[mb].status = ADP_MB_MAD_ZERO; // nothing to normalise: treated like a failed minibatch
        }
        for (int j = 0; j < per; j++) gh[threadIdx.x * per + j] = 0;
        return;
    }
    unsigned long long cum = part[threadIdx.x];
    for (int j = 0; j < per; j++) {
        if (st.krem >= cum && st.krem < cum + loc[j]) { s_bin = threadIdx.x * per + j; s_before = cum; }
        cum += loc[j];
    }
    __syncthreads();
    const int bin = s_bin;
    const unsigned long long before = s_before;
    if (PASS < 2) {
        if (threadIdx.x == 0) {
            mbs[mb].prefix = (PASS == 0) ? (uint32_t)bin : ((st.prefix << 11) | (uint32_t)bin);
            mbs[mb].krem = st.krem - before;
        }
    } else {
        // final: exact key; lower median if the count is even
        __shared__ int s_lowbin;
        if (threadIdx.x == 0) s_lowbin = -1;
        __syncthreads();
        // largest non-empty bin below `bin` inside this bucket
        int cand = -1;
        for (int j = 0; j < per; j++) { int b = threadIdx.x * per + j; if (b < bin && loc[j]) cand = b; }
        if (cand >= 0) atomicMax(&s_lowbin, cand);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t key = (st.prefix << 10) | (uint32_t)bin;
            float v = key2f(key);
            float res = v;
            if ((st.n_valid & 1ull) == 0) {
                unsigned long long rank_in_bin = st.krem - before;
                float v0 = v;
                if (rank_in_bin == 0) {
                    uint32_t k0 = gbelow[mb];
                    if (s_lowbin >= 0) { uint32_t ka = (st.prefix << 10) | (uint32_t)s_lowbin; if (ka > k0) k0 = ka; }
                    v0 = key2f(k0);
                }
                res = (v0 + v) / 2.0f;
            }
            if (mode == 0) {
                mbs[mb].med = res;
            } else {
                mbs[mb].mad = res;
                double dmed = (double)st.med, dmad = (double)res;
                mbs[mb].lo = (float)(dmed - dmad * thresh);
                mbs[mb].hi = (float)(dmed + dmad * thresh);
                if (res == 0.0f) mbs[mb].status = ADP_MB_MAD_ZERO;
            }
            mbs[mb].prefix = 0; mbs[mb].krem = 0;
        }
    }
    for (int j = 0; j < per; j++) gh[threadIdx.x * per + j] = 0;
    if (PASS == 2) { for (int i = threadIdx.x + 1024; i < 2048; i += 256) gh[i] = 0; }
    if (threadIdx.x == 0) gbelow[mb] = 0;
}
