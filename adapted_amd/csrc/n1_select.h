// n1_select.h -- N1: exact batch-global nanmedian / MAD of batch[:, :T] per minibatch.
//
// Replaces normalize_signal/med_mad (reference adapted/detect/normalize.py:15-63) as called
// with the whole 2-D minibatch at adapted/detect/combined.py:128-132: ONE median and ONE MAD
// per minibatch, computed exactly (k-th order statistics; even count -> float32 mean of the
// two middle values, like numpy).
//
// Method: radix select on order-preserving uint32 keys in TWO full HBM passes per statistic
// (four in total) instead of the textbook three:
//   guess   the median of a row sample (every row_step-th read; two cheap histogram passes over the
//           sample) centres a window of 2^21 consecutive keys (16 pA wide near 100 pA);
//   pass 1  over all samples: count the keys below the window, histogram the keys inside it in
//           2048 buckets of 1024 keys, count the valid samples.  If the wanted rank falls inside
//           the window (verified exactly from the counts) its bucket is known;
//   pass 2  histogram the 1024 keys of that bucket and track the largest key below it (needed for
//           the lower median of an even count) -> exact key.
// A missed guess is detected, never trusted: the classic aligned 11/11/10-bit path then runs
// (three passes).  Histograms are per-block in LDS (float4 loads, per-lane run-length aggregation
// of equal digits against same-address LDS atomics) and flushed to one global histogram per
// minibatch.
#pragma once
#include "common.h"

#define N1_BINS 2048
#define N1_THREADS 256
#define N1_WIN (1u << 21)

enum { N1_ALWAYS = 0, N1_IF_BAD = 1 };

// transform: mode 0 -> x ; mode 1 -> |x - med| (float32, as numpy computes np.abs(signal - med))
static __device__ __forceinline__ float n1_xform(float x, int mode, float med) { return mode ? fabsf(x - med) : x; }

struct N1Acc { uint32_t below, d_prev, run; uint32_t nvalid, nbelow; };

template <int PASS>
static __device__ __forceinline__ void n1_account(float x, int mode, float med, uint32_t kbase, LDS uint32_t *hist, N1Acc &a)
{
    float v = n1_xform(x, mode, med);
    if (v != v) return;
    uint32_t key = f2key(v);
    uint32_t digit;
    if (PASS == 0) digit = key >> 21;
    else if (PASS == 1) {
        a.nvalid++;
        uint32_t d = key - kbase;
        if (d >= N1_WIN) { if (key < kbase) { a.nbelow++; if (key > a.below) a.below = key; } return; }
        digit = d >> 10;
    } else {
        uint32_t d = key - kbase;
        if (d >= 1024u) { if (key < kbase && key > a.below) a.below = key; return; }
        digit = d;
    }
    // run-length aggregation inside the lane
    if (digit == a.d_prev) a.run++;
    else {
        if (a.run) __hip_atomic_fetch_add(&hist[a.d_prev], a.run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        a.d_prev = digit; a.run = 1;
    }
}

// grid = (blocks_per_minibatch, n_minibatch); block = N1_THREADS.  Rows r0 + j*row_step of the minibatch.
template <int PASS>
__global__ void __launch_bounds__(N1_THREADS) k_n1_hist(const float *__restrict__ sig, int n_reads, int m, int T, int mbsize,
                                                         int mode, const MbState *__restrict__ mbs, uint32_t *__restrict__ ghist,
                                                         uint32_t *__restrict__ gbelow, unsigned long long *__restrict__ gcnt,
                                                         int row_step, int when)
{
    __shared__ uint32_t hist_[N1_BINS];
    __shared__ uint32_t sbelow;
    LDS uint32_t *hist = (LDS uint32_t *)hist_;
    const int mb = blockIdx.y;
    const MbState st = mbs[mb];
    if (st.status != ADP_MB_OK) return;
    if (when == N1_IF_BAD && !st.bad) return;
    for (int i = threadIdx.x; i < N1_BINS; i += N1_THREADS) hist[i] = 0;
    if (threadIdx.x == 0) sbelow = 0;
    __syncthreads();
    const int r0 = mb * mbsize;
    const int r1 = min(n_reads, r0 + mbsize);
    const float med = st.med;
    const uint32_t kbase = st.kbase;
    N1Acc a; a.below = 0; a.d_prev = 0xffffffffu; a.run = 0; a.nvalid = 0; a.nbelow = 0;
    const bool vec = ((m & 3) == 0) && ((reinterpret_cast<uintptr_t>(sig) & 15) == 0);
    for (long long r = r0 + (long long)blockIdx.x * row_step; r < r1; r += (long long)gridDim.x * row_step) {
        const float *row = sig + (size_t)r * m;
        if (vec) {
            const int T4 = T >> 2;
            const float4 *row4 = reinterpret_cast<const float4 *>(row);
            for (int i = threadIdx.x; i < T4; i += N1_THREADS) {
                float4 v = row4[i];
                n1_account<PASS>(v.x, mode, med, kbase, hist, a);
                n1_account<PASS>(v.y, mode, med, kbase, hist, a);
                n1_account<PASS>(v.z, mode, med, kbase, hist, a);
                n1_account<PASS>(v.w, mode, med, kbase, hist, a);
            }
            for (int i = (T4 << 2) + threadIdx.x; i < T; i += N1_THREADS)
                n1_account<PASS>(row[i], mode, med, kbase, hist, a);
        } else {
            for (int i = threadIdx.x; i < T; i += N1_THREADS)
                n1_account<PASS>(row[i], mode, med, kbase, hist, a);
        }
    }
    if (a.run) __hip_atomic_fetch_add(&hist[a.d_prev], a.run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (PASS > 0) {
        uint32_t below = wave_max(a.below);
        if (lane_id() == 0 && below) atomicMax(&sbelow, below);
    }
    if (PASS == 1) {
        unsigned long long nv = (unsigned long long)wave_sum((int)a.nvalid), nb = (unsigned long long)wave_sum((int)a.nbelow);
        if (lane_id() == 0) {
            if (nv) atomicAdd(&gcnt[2 * mb], nv);
            if (nb) atomicAdd(&gcnt[2 * mb + 1], nb);
        }
    }
    __syncthreads();
    uint32_t *gh = ghist + (size_t)mb * N1_BINS;
    for (int i = threadIdx.x; i < N1_BINS; i += N1_THREADS) {
        uint32_t c = hist[i];
        if (c) atomicAdd(&gh[i], c);
    }
    if (PASS > 0 && threadIdx.x == 0 && sbelow) atomicMax(&gbelow[mb], sbelow);
}

// block-wide: total of the histogram and the bin holding rank k (bin = -1 if k >= total)
static __device__ void n1_find(const uint32_t *gh, int nb, unsigned long long k, unsigned long long *part, int *s_bin,
                               unsigned long long *s_before, unsigned long long *s_total, uint32_t *loc)
{
    const int per = nb / 256;
    unsigned long long s = 0;
    for (int j = 0; j < per; j++) { loc[j] = gh[threadIdx.x * per + j]; s += loc[j]; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 256; i++) { unsigned long long t = part[i]; part[i] = tot; tot += t; }
        *s_total = tot; *s_bin = -1; *s_before = 0;
    }
    __syncthreads();
    unsigned long long cum = part[threadIdx.x];
    for (int j = 0; j < per; j++) {
        if (k >= cum && k < cum + loc[j]) { *s_bin = threadIdx.x * per + j; *s_before = cum; }
        cum += loc[j];
    }
    __syncthreads();
}

enum { N1_SAMPLE = 0, N1_FULL = 1, N1_FALLBACK = 2 };

// One block per minibatch.  PASS 0: aligned top-11-bit bucket of the rank (sample guess or fallback).
// PASS 1: KIND sample -> centre the 2^21-key window on the sample median; KIND full -> verify the guess
// against the exact counts and descend; KIND fallback -> descend inside the aligned bucket.
// PASS 2: exact key, lower median of an even count, write med (mode 0) or mad + clip bounds (mode 1).
template <int PASS, int KIND>
__global__ void __launch_bounds__(256) k_n1_pick(MbState *__restrict__ mbs, uint32_t *__restrict__ ghist,
                                                 uint32_t *__restrict__ gbelow, unsigned long long *__restrict__ gcnt,
                                                 int mode, double thresh)
{
    __shared__ unsigned long long part[256];
    __shared__ int s_bin, s_lowbin;
    __shared__ unsigned long long s_before, s_total;
    const int mb = blockIdx.x;
    MbState st = mbs[mb];
    if (st.status != ADP_MB_OK) return;
    if (KIND == N1_FALLBACK && !st.bad) return;
    uint32_t *gh = ghist + (size_t)mb * N1_BINS;
    const int nb = (PASS == 2) ? 1024 : 2048;
    const int per = nb / 256;
    uint32_t loc[8];
    unsigned long long k;
    if (PASS == 0) {
        // rank of the upper median of whatever was histogrammed (the sample, or everything)
        unsigned long long s = 0;
        for (int j = 0; j < per; j++) s += gh[threadIdx.x * per + j];
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long t = 0; for (int i = 0; i < 256; i++) t += part[i]; s_total = t; }
        __syncthreads();
        k = s_total / 2;
        __syncthreads();
    } else if (PASS == 1 && KIND == N1_FULL) {
        st.n_valid = gcnt[2 * mb];
        st.c_below = gcnt[2 * mb + 1];
        k = st.n_valid / 2; // absolute rank; made window-relative below
    } else {
        k = st.krem;
    }
    bool miss = false;
    unsigned long long krel = k;
    if (PASS == 1 && KIND == N1_FULL) {
        if (k < st.c_below) { miss = true; krel = 0; } else krel = k - st.c_below;
    }
    n1_find(gh, nb, krel, part, &s_bin, &s_before, &s_total, loc);
    const int bin = s_bin;
    const unsigned long long before = s_before;
    if (bin < 0) miss = true;

    if (PASS == 1 && KIND == N1_FULL && st.n_valid == 0) {
        if (threadIdx.x == 0) {
            mbs[mb].med = __builtin_nanf(""); mbs[mb].mad = __builtin_nanf("");
            mbs[mb].status = ADP_MB_MAD_ZERO; // nothing to normalise: treated like a failed minibatch
        }
    } else if (PASS == 0) {
        if (threadIdx.x == 0) {
            if (KIND == N1_FALLBACK) mbs[mb].n_valid = s_total;
            mbs[mb].kbase = (bin < 0) ? 0u : ((uint32_t)bin << 21);
            mbs[mb].krem = (bin < 0) ? 0ull : k - before;
        }
    } else if (PASS == 1) {
        if (threadIdx.x == 0) {
            if (KIND == N1_SAMPLE) {
                // centre of the sample median's 1024-key bucket -> window start
                unsigned long long c = (unsigned long long)st.kbase + (bin < 0 ? (N1_WIN >> 1) : (((unsigned long long)bin << 10) + 512ull));
                unsigned long long lo = c > (N1_WIN >> 1) ? c - (N1_WIN >> 1) : 0ull;
                if (lo > 0xFFFFFFFFull - N1_WIN) lo = 0xFFFFFFFFull - N1_WIN;
                mbs[mb].kbase = (uint32_t)lo;
                mbs[mb].bad = 0;
            } else if (KIND == N1_FULL) {
                mbs[mb].n_valid = st.n_valid;
                mbs[mb].c_below = st.c_below;
                if (miss) mbs[mb].bad = 1;
                else { mbs[mb].bad = 0; mbs[mb].kbase = st.kbase + ((uint32_t)bin << 10); mbs[mb].krem = krel - before; }
            } else {
                mbs[mb].bad = 0; // the aligned bucket always holds the rank
                mbs[mb].kbase = st.kbase + ((uint32_t)(bin < 0 ? 0 : bin) << 10);
                mbs[mb].krem = k - before;
            }
        }
    } else {
        // final: exact key; lower median if the count is even
        if (threadIdx.x == 0) s_lowbin = -1;
        __syncthreads();
        int cand = -1; // largest non-empty bin below `bin` inside this bucket
        for (int j = 0; j < per; j++) { int b = threadIdx.x * per + j; if (b < bin && loc[j]) cand = b; }
        if (cand >= 0) atomicMax(&s_lowbin, cand);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t key = st.kbase + (uint32_t)(bin < 0 ? 0 : bin);
            float v = key2f(key);
            float res = v;
            if ((st.n_valid & 1ull) == 0) {
                unsigned long long rank_in_bin = k - before;
                float v0 = v;
                if (rank_in_bin == 0) {
                    uint32_t k0 = gbelow[mb];
                    if (s_lowbin >= 0) { uint32_t ka = st.kbase + (uint32_t)s_lowbin; if (ka > k0) k0 = ka; }
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
            mbs[mb].kbase = 0; mbs[mb].krem = 0; mbs[mb].bad = 0;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N1_BINS; i += 256) gh[i] = 0;
    if (threadIdx.x == 0) {
        gbelow[mb] = 0;
        if (PASS == 1) { gcnt[2 * mb] = 0; gcnt[2 * mb + 1] = 0; } // also the sample's / fallback's counts
    }
}
