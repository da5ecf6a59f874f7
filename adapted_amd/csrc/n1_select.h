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

#define N1_CB_LDS 4096          // per-block staging of bracket keys (flushed after every row)
#define N1_CB_CAP (1u << 22)    // bracket keys kept per minibatch
// half width of the sampled rank brackets in standard deviations of an independent sample (round 3: 4.5 -> 6; the sample is
// clustered -- pieces of 96 consecutive samples -- and 1 of 96 Pareto-length minibatches missed its median bracket at 4.5, which
// costs that minibatch three more passes; the wider bracket copies a third more of the ~1.5 % of samples it copies)
#ifndef N1_BRACKET_SIGMAS
#define N1_BRACKET_SIGMAS 6.0
#endif
#define N1_NCNT 8               // u64 counters per minibatch: n_valid, below window, below bracket, bracket count, overflow

enum { N1_ALWAYS = 0, N1_IF_BAD = 1, N1_IF_NOT_DONE = 2 };

extern __device__ unsigned long long g_dbg[ADP_NDBG];

// transform: mode 0 -> x ; mode 1 -> |x - med| (float32, as numpy computes np.abs(signal - med))
static __device__ __forceinline__ float n1_xform(float x, int mode, float med) { return mode ? fabsf(x - med) : x; }

struct N1Acc { uint32_t below, d_prev, run; uint32_t nvalid, nbelow; uint32_t below2, nbelow2; };

template <int PASS>
static __device__ __forceinline__ void n1_account(float x, int mode, float med, uint32_t kbase, LDS uint32_t *hist, N1Acc &a,
                                                   uint32_t cklo, uint32_t ckw, LDS uint32_t *cb, LDS uint32_t *cb_cnt)
{
    float v = n1_xform(x, mode, med);
    if (v != v) return;
    uint32_t key = f2key(v);
    uint32_t digit;
    if (PASS == 0) digit = key >> 21;
    else if (PASS == 1) {
        a.nvalid++;
        if (ckw) { // narrow bracket around the sampled rank: copy its keys out, count / track what lies below
            uint32_t dd = key - cklo;
            const bool lower = key < cklo;
            a.nbelow2 += lower ? 1u : 0u;
            const uint32_t cand = lower ? key : 0u;
            a.below2 = cand > a.below2 ? cand : a.below2;
            if (dd < ckw) { // ~0.3 % of the samples
                uint32_t slot = __hip_atomic_fetch_add(cb_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (slot < N1_CB_LDS) cb[slot] = key;
            }
        }
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
template <int PASS, class SIG>
__global__ void __launch_bounds__(N1_THREADS) k_n1_hist(SIG sig, int n_reads, int m, int T, int mbsize,
                                                         int mode, const MbState *__restrict__ mbs, uint32_t *__restrict__ ghist,
                                                         uint32_t *__restrict__ gbelow, unsigned long long *__restrict__ gcnt,
                                                         int row_step, int when, uint32_t *__restrict__ cbuf, int collect, int col_div, int pdiv = 8,
                                                         const int32_t *__restrict__ full_len = nullptr)
{
    __shared__ __attribute__((aligned(16))) uint32_t hist_[N1_BINS];
    __shared__ uint32_t sbelow, sbelow2, cb_cnt_, cb_base;
    __shared__ __attribute__((aligned(16))) uint32_t cb_[PASS == 1 ? N1_CB_LDS : 1];
    LDS uint32_t *hist = (LDS uint32_t *)hist_;
    LDS uint32_t *cb = (LDS uint32_t *)cb_;
    LDS uint32_t *cb_cnt = (LDS uint32_t *)&cb_cnt_;
    const int mb = blockIdx.y;
    const MbState st = mbs[mb];
    if (st.status != ADP_MB_OK || st.fused) return;
    if (when == N1_IF_BAD && !st.bad) return;
    if (when == N1_IF_NOT_DONE && st.done) return;
    const uint32_t cklo = st.cklo, ckw = (PASS == 1 && collect) ? st.ckw : 0u;
    for (int i = threadIdx.x; i < N1_BINS; i += N1_THREADS) hist[i] = 0;
    if (threadIdx.x == 0) { sbelow = 0; sbelow2 = 0; cb_cnt_ = 0; }
    __syncthreads();
    const int r0 = mb * mbsize;
    const int r1 = min(n_reads, r0 + mbsize);
    const float med = st.med;
    const uint32_t kbase = st.kbase;
    N1Acc a; a.below = 0; a.d_prev = 0xffffffffu; a.run = 0; a.nvalid = 0; a.nbelow = 0; a.below2 = 0; a.nbelow2 = 0;
    const bool vec = sig.vec_ok(m);
    for (long long r = r0 + (long long)blockIdx.x * row_step; r < r1; r += (long long)gridDim.x * row_step) {
        const typename SIG::Row row = sig.row(r, m);
        // ADP_TAILS_NAN: everything from the read's end on is NaN padding, which counts for nothing here
        int Te = T;
        if (full_len) { const int fl = full_len[r]; Te = fl < T ? (fl > 0 ? fl : 0) : T; }
        if (col_div > 1) {
            // sampling pass: col_div contiguous pieces of T / (col_div * pdiv) samples, one in every col_div-th of the window
            // (reads of very different lengths still contribute from every part they have -- with every read sampled, the
            // sample's mix of reads is the minibatch's own), the position inside the part rotating from row to row
            const int Rg = (T / col_div) & ~3, Tp = (T / (col_div * pdiv)) & ~3;
            const int rot = (int)(((r - r0) / row_step) % pdiv);
            if (vec) { // all pieces of the row as one index space: (piece, float4 inside it)
                const int p4 = Tp >> 2, tot4 = col_div * p4;
                const int rg4 = Rg >> 2, off4 = (rot * Tp) >> 2;
                for (int i = threadIdx.x; i < tot4; i += N1_THREADS) {
                    const int pc = i / p4, j = i - pc * p4;
                    if (4 * (pc * rg4 + off4 + j) >= Te) continue;
                    float4 v = row.f4(pc * rg4 + off4 + j);
                    n1_account<PASS>(v.x, mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
                    n1_account<PASS>(v.y, mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
                    n1_account<PASS>(v.z, mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
                    n1_account<PASS>(v.w, mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
                }
            } else {
                const int tot = col_div * Tp;
                for (int i = threadIdx.x; i < tot; i += N1_THREADS) {
                    const int pc = i / Tp, j = i - pc * Tp;
                    if (pc * Rg + rot * Tp + j >= Te) continue;
                    n1_account<PASS>(row[(size_t)pc * Rg + (size_t)rot * Tp + j], mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
                }
            }
        } else if (vec) {
            const int T4 = Te >> 2;
            for (int i = threadIdx.x; i < T4; i += N1_THREADS) {
                float4 v = row.f4(i);
                n1_account<PASS>(v.x, mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
                n1_account<PASS>(v.y, mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
                n1_account<PASS>(v.z, mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
                n1_account<PASS>(v.w, mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
            }
            for (int i = (T4 << 2) + threadIdx.x; i < Te; i += N1_THREADS)
                n1_account<PASS>(row[i], mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
        } else {
            for (int i = threadIdx.x; i < Te; i += N1_THREADS)
                n1_account<PASS>(row[i], mode, med, kbase, hist, a, cklo, ckw, cb, cb_cnt);
        }
        if (PASS == 1 && ckw) { // hand the row's bracket keys to the minibatch buffer: one global atomic per row and block
            __syncthreads();
            const uint32_t cnt = cb_cnt_;
            if (threadIdx.x == 0) {
                if (cnt > N1_CB_LDS) { atomicAdd(&gcnt[N1_NCNT * mb + 4], 1ull); cb_base = 0xffffffffu; }
                else cb_base = cnt ? (uint32_t)atomicAdd(&gcnt[N1_NCNT * mb + 3], (unsigned long long)cnt) : 0u;
            }
            __syncthreads();
            const uint32_t base = cb_base;
            if (base != 0xffffffffu) {
                uint32_t *dst = cbuf + (size_t)mb * N1_CB_CAP;
                for (uint32_t i = threadIdx.x; i < cnt; i += N1_THREADS) if (base + i < N1_CB_CAP) dst[base + i] = cb[i];
            }
            __syncthreads();
            if (threadIdx.x == 0) cb_cnt_ = 0;
            __syncthreads();
        }
    }
    if (a.run) __hip_atomic_fetch_add(&hist[a.d_prev], a.run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (PASS > 0) {
        uint32_t below = wave_max(a.below);
        if (lane_id() == 0 && below) atomicMax(&sbelow, below);
    }
    if (PASS == 1) {
        unsigned long long nv = (unsigned long long)wave_sum((int)a.nvalid), nb = (unsigned long long)wave_sum((int)a.nbelow);
        unsigned long long nb2 = (unsigned long long)wave_sum((int)a.nbelow2);
        uint32_t b2 = wave_max(a.below2);
        if (lane_id() == 0) {
            if (nv) atomicAdd(&gcnt[N1_NCNT * mb], nv);
            if (nb) atomicAdd(&gcnt[N1_NCNT * mb + 1], nb);
            if (nb2) atomicAdd(&gcnt[N1_NCNT * mb + 2], nb2);
            if (b2) atomicMax(&sbelow2, b2);
        }
    }
    __syncthreads();
    uint32_t *gh = ghist + (size_t)mb * N1_BINS;
    for (int i = threadIdx.x; i < N1_BINS; i += N1_THREADS) {
        uint32_t c = hist[i];
        if (c) atomicAdd(&gh[i], c);
    }
    if (PASS > 0 && threadIdx.x == 0 && sbelow) atomicMax(&gbelow[2 * mb], sbelow);
    if (PASS == 1 && threadIdx.x == 0 && sbelow2) atomicMax(&gbelow[2 * mb + 1], sbelow2);
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
    __shared__ __attribute__((aligned(16))) unsigned long long part[256];
    __shared__ int s_bin, s_lowbin;
    __shared__ unsigned long long s_before, s_total;
    const int mb = blockIdx.x;
    MbState st = mbs[mb];
    if (st.status != ADP_MB_OK || st.fused) return;
    if (KIND == N1_FALLBACK && !st.bad) return;
    if (PASS == 2 && st.done) return; // finished from the copied bracket (k_n1_finish)
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
    } else if (PASS == 1 && (KIND == N1_FULL || KIND == N1_SAMPLE)) {
        st.n_valid = gcnt[N1_NCNT * mb];       // (of the sample for KIND sample)
        st.c_below = gcnt[N1_NCNT * mb + 1];
        k = st.n_valid / 2; // absolute rank; made window-relative below
    } else {
        k = st.krem;
    }
    bool miss = false;
    unsigned long long krel = k;
    if (PASS == 1 && (KIND == N1_FULL || KIND == N1_SAMPLE)) {
        if (k < st.c_below) { miss = true; krel = 0; } else krel = k - st.c_below;
    }
    n1_find(gh, nb, krel, part, &s_bin, &s_before, &s_total, loc);
    const int bin = s_bin;
    const unsigned long long before = s_before;
    if (bin < 0) miss = true;
    if (PASS == 1 && KIND == N1_SAMPLE) {
        // bracket: the buckets holding the sample ranks krem -/+ D, D = 8 standard deviations of the rank of the
        // true median inside an independent sample of M_s (binomial) -- reads differ, so the sample is clustered
        // and the bracket is only a good bet; whether it held is verified exactly after pass 1
        const unsigned long long tot = s_total;
        const unsigned long long Ms = st.n_valid; // sample size
        // (with every read in the sample the clustering is that of positions inside a read only; a bracket that misses
        // costs the multi-pass selection, never the result)
        // ... as wide as the lists allow: the fused pass copies about 2 D x 32 samples (the sample is a 32nd of the minibatch)
        // into a median list of N1_CB_CAP / 4 entries; at the 200 k window that caps D near 4.6 sigma
        unsigned long long D = (unsigned long long)(N1_BRACKET_SIGMAS * sqrt((double)Ms)) + 16ull;
        { const unsigned long long dcap = (unsigned long long)(N1_CB_CAP / 4) * 7ull / 10ull / 64ull; if (D > dcap) D = dcap; }
        const bool cut = miss || krel < D || krel + D >= tot; // the bracket would be cut off by the window: no bracket
        const unsigned long long rlo = krel > D ? krel - D : 0ull;
        unsigned long long rhi = krel + D;
        if (tot && rhi > tot - 1) rhi = tot - 1;
        __syncthreads();
        n1_find(gh, nb, rlo, part, &s_bin, &s_before, &s_total, loc);
        const int blo = s_bin;
        __syncthreads();
        n1_find(gh, nb, rhi, part, &s_bin, &s_before, &s_total, loc);
        const int bhi = s_bin;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (tot && !cut && blo >= 0 && bhi >= blo) { mbs[mb].cklo = st.kbase + ((uint32_t)blo << 10); mbs[mb].ckw = (uint32_t)(bhi + 1 - blo) << 10; }
            else { mbs[mb].cklo = 0; mbs[mb].ckw = 0; }
        }
    }

    if (PASS == 1 && KIND == N1_FULL && st.n_valid == 0) {
        if (threadIdx.x == 0) {
            mbs[mb].med = __builtin_nanf(""); mbs[mb].mad = __builtin_nanf("");
            mbs[mb].status = ADP_MB_MAD_ZERO; // nothing to normalise: treated like a failed minibatch
        }
    } else if (PASS == 0) {
        if (threadIdx.x == 0) {
            if (KIND == N1_FALLBACK) mbs[mb].n_valid = s_total;
            if (KIND == N1_SAMPLE) {
                // centre the sample's second pass on the median interpolated inside its top-11-bit bucket, so that the
                // bracket around it is not cut off by a bucket edge
                mbs[mb].done = 0; mbs[mb].ckw = 0;
                unsigned long long c = 0;
                if (bin >= 0) {
                    unsigned long long cnt = gh[bin];
                    double frac = cnt ? (double)(k - before) / (double)cnt : 0.5;
                    c = ((unsigned long long)bin << 21) + (unsigned long long)(frac * 2097152.0);
                }
                unsigned long long lo = c > (N1_WIN >> 1) ? c - (N1_WIN >> 1) : 0ull;
                lo &= ~1023ull;
                if (lo > 0xFFFFFFFFull - N1_WIN) lo = (0xFFFFFFFFull - N1_WIN) & ~1023ull;
                mbs[mb].kbase = (uint32_t)lo;
                mbs[mb].krem = 0;
            } else {
                mbs[mb].kbase = (bin < 0) ? 0u : ((uint32_t)bin << 21);
                mbs[mb].krem = (bin < 0) ? 0ull : k - before;
            }
        }
    } else if (PASS == 1) {
        if (threadIdx.x == 0) {
            if (KIND == N1_SAMPLE) {
                // the sample median's 1024-key bucket -> centre of the full pass's window (same 1024-key grid)
                unsigned long long c = (unsigned long long)st.kbase + (bin < 0 ? (N1_WIN >> 1) : ((unsigned long long)bin << 10));
                unsigned long long lo = c > (N1_WIN >> 1) ? c - (N1_WIN >> 1) : 0ull;
                lo &= ~1023ull;
                if (lo > 0xFFFFFFFFull - N1_WIN) lo = (0xFFFFFFFFull - N1_WIN) & ~1023ull;
                mbs[mb].kbase = (uint32_t)lo;
                mbs[mb].bad = 0;
            } else if (KIND == N1_FULL) {
                // was the rank inside the copied bracket?  (below-bracket count, bracket count, overflow flag)
                const unsigned long long nb2 = gcnt[N1_NCNT * mb + 2], nc = gcnt[N1_NCNT * mb + 3], ovf = gcnt[N1_NCNT * mb + 4];
                if (st.ckw && !ovf && nc <= N1_CB_CAP && k >= nb2 && k < nb2 + nc) {
                    mbs[mb].done = 2;              // k_n1_finish selects inside the bracket
                    mbs[mb].c_below = k - nb2;     // rank inside the bracket
                    mbs[mb].n_valid = st.n_valid;
                    mbs[mb].bad = 0;
                } else {
                mbs[mb].n_valid = st.n_valid;
                mbs[mb].c_below = st.c_below;
                if (miss) mbs[mb].bad = 1;
                else { mbs[mb].bad = 0; mbs[mb].kbase = st.kbase + ((uint32_t)bin << 10); mbs[mb].krem = krel - before; }
                }
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
                    uint32_t k0 = gbelow[2 * mb];
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
        gbelow[2 * mb] = 0;
        if (!(PASS == 1 && KIND == N1_FULL)) gbelow[2 * mb + 1] = 0; // the finish kernel still needs the bracket's "below" key
        if (PASS == 1 && KIND != N1_FULL) for (int i = 0; i < N1_NCNT; i++) gcnt[N1_NCNT * mb + i] = 0; // sample's / fallback's counts
        if (PASS == 1 && KIND == N1_FULL) { gcnt[N1_NCNT * mb] = 0; gcnt[N1_NCNT * mb + 1] = 0; } // [2..4] are consumed by k_n1_finish
    }
}

// One block per minibatch: when pass 1 proved that the wanted rank lies inside the copied bracket, select it
// there (MSB-first 8-bit radix over key - min, LDS histogram) and finish the statistic: pass 2 is skipped.
__global__ void __launch_bounds__(1024) k_n1_finish(MbState *__restrict__ mbs, const uint32_t *__restrict__ cbuf,
                                                    unsigned long long *__restrict__ gcnt, uint32_t *__restrict__ gbelow, int mode,
                                                    double thresh)
{
    __shared__ __attribute__((aligned(16))) uint32_t hist[256];
    __shared__ uint32_t s_min, s_max, s_below, s_prefix, s_k;
    __shared__ int s_low, s_lastw;
    const int mb = blockIdx.x;
    const MbState st = mbs[mb];
    const int tid = threadIdx.x;
    if (st.status != ADP_MB_OK || st.done != 2 || st.fused) {
        if (tid == 0) { gcnt[N1_NCNT * mb + 2] = 0; gcnt[N1_NCNT * mb + 3] = 0; gcnt[N1_NCNT * mb + 4] = 0; gbelow[2 * mb + 1] = 0; }
        return;
    }
    const uint32_t n = (uint32_t)gcnt[N1_NCNT * mb + 3];
    const uint32_t *keys = cbuf + (size_t)mb * N1_CB_CAP;
    if (tid == 0) { s_min = 0xffffffffu; s_max = 0; s_below = 0; s_prefix = 0; s_k = (uint32_t)st.c_below; s_low = -1; }
    __syncthreads();
    uint32_t mn = 0xffffffffu, mx = 0;
    for (uint32_t i = tid; i < n; i += 1024) { uint32_t key = keys[i]; mn = key < mn ? key : mn; mx = key > mx ? key : mx; }
    mn = wave_min(mn); mx = wave_max(mx);
    if (lane_id() == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
    __syncthreads();
    mn = s_min; mx = s_max;
    const uint32_t span = mx - mn;
    int rb = span ? 32 - __clz(span) : 0; // bits still unresolved of d = key - mn
    while (rb > 0) {
        const int w = rb < 8 ? rb : 8;
        const int shift = rb - w;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        uint32_t below = 0;
        for (uint32_t i = tid; i < n; i += 1024) {
            uint32_t d = keys[i] - mn;
            uint32_t top = (rb >= 32) ? 0u : (d >> rb);
            if (top == prefix) atomicAdd(&hist[(d >> shift) & ((1u << w) - 1u)], 1u);
            else if (shift == 0 && top < prefix && d + 1u > below) below = d + 1u; // (+1: 0 means "none")
        }
        if (shift == 0) { below = wave_max(below); if (lane_id() == 0 && below) atomicMax(&s_below, below); }
        __syncthreads();
        if (tid == 0) {
            uint32_t k = s_k, cum = 0; int bin = (1 << w) - 1, low = -1;
            for (int b = 0; b < (1 << w); b++) {
                uint32_t h = hist[b];
                if (k < cum + h) { bin = b; break; }
                if (h) low = b;
                cum += h;
            }
            s_k = k - cum;
            s_prefix = (prefix << w) | (uint32_t)bin;
            if (shift == 0) { s_low = low; s_lastw = w; }
        }
        __syncthreads();
        rb = shift;
    }
    if (tid == 0) {
        const uint32_t dsel = s_prefix;
        const float v = key2f(mn + dsel);
        float res = v;
        if ((st.n_valid & 1ull) == 0) {
            float v0 = v;
            if (s_k == 0) { // the lower median is the largest key below the selected one
                uint32_t k0 = gbelow[2 * mb + 1]; // below the bracket (pass 1)
                if (s_below) { uint32_t ka = mn + (s_below - 1u); if (ka > k0) k0 = ka; }
                if (span && s_low >= 0) { uint32_t ka = mn + ((dsel & ~((1u << s_lastw) - 1u)) | (uint32_t)s_low); if (ka > k0) k0 = ka; }
                v0 = key2f(k0);
            }
            res = (v0 + v) / 2.0f;
        }
        if (mode == 0) mbs[mb].med = res;
        else {
            mbs[mb].mad = res;
            double dmed = (double)st.med, dmad = (double)res;
            mbs[mb].lo = (float)(dmed - dmad * thresh);
            mbs[mb].hi = (float)(dmed + dmad * thresh);
            if (res == 0.0f) mbs[mb].status = ADP_MB_MAD_ZERO;
        }
        mbs[mb].done = 1; mbs[mb].kbase = 0; mbs[mb].krem = 0; mbs[mb].bad = 0; mbs[mb].ckw = 0;
        gcnt[N1_NCNT * mb + 2] = 0; gcnt[N1_NCNT * mb + 3] = 0; gcnt[N1_NCNT * mb + 4] = 0; gbelow[2 * mb + 1] = 0;
    }
}
