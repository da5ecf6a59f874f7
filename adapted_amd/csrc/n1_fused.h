// n1_fused.h -- N1 in ONE pass over the minibatch: median and MAD of batch[:, :T] together.
//
// n1_select.h needs one full HBM pass per statistic because the MAD's keys |x - med| depend on the median.
// Here both are settled from a single pass, exactly, for minibatches big enough to sample:
//
//   sample   (the same two cheap histogram passes over the row/column sample as n1_select.h, run for both
//            statistics up front) gives   med_s        a median estimate,
//                                         [A0, A1)     a bracket of u = x - med_s that should hold the true median,
//                                         [D0, D1]     a bracket of t = |x - med_s| that should hold the true MAD,
//                                                      widened by the width of the median bracket;
//   pass     over all samples: count the valid ones, those with u < A0 ("below") and those with t < D0
//            ("inner"); copy out every x with u in [A0, A1) or t in [D0, D1]  (about 1.5 % of the samples);
//   finish   one block per minibatch: the median is the (n/2 - below)-th copied value of the first bracket if
//            that rank falls inside it (exact: u is monotone in x).  With the exact median known, every sample
//            whose v = |x - med| lies strictly between
//                  vlo = D0 + |med - med_s|  (rounded up)   and   vhi = D1 - |med - med_s|  (rounded down)
//            was copied (inner samples have v <= vlo, outer ones v >= vhi, because float32 subtraction is
//            monotone), so the MAD is the (n/2 - inner - #{copied: v <= vlo})-th of the copied v inside that
//            zone, if the rank falls there.
// Anything that does not verify (rank outside a bracket, staging overflow, degenerate sample) leaves the
// minibatch to n1_select.h's passes, which run afterwards for exactly those minibatches.
#pragma once
#include "n1_select.h"

#define N1F_LDS_M 768             // per-block staging (floats) of the median bracket, flushed after every quarter row
#define N1F_LDS_B 2304            // ... of the MAD band
#define N1F_MCAP (N1_CB_CAP / 4)  // copied samples kept per minibatch: median bracket ...
#define N1F_BCAP (N1_CB_CAP - N1F_MCAP) // ... and MAD band (the two lists share the minibatch's slice of cbuf)
#define N1F_BINS 32768            // finish: LDS counting histogram (128 KB)

struct N1Fused {
    float med_s, A0, A1, D0, D1, eps;
    int32_t ok, wide_ok;
    float D0w, D1w; // the MAD band widened by another eps on either side: used when the data have heavy keys (see k_n1_heavy_pick)
};
// counters per minibatch (u64): 0 valid, 1 below (u < A0), 2 inner (t < D0), 3 copied (median bracket), 4 staging overflow,
// 5 copied (MAD band)
#define N1F_NCNT 8

// ---- heavy keys: values shared by a large part of the copied samples --------------------------------------------
// Calibrated int16 ADC data lie on one grid (pA = scale * (adc + offset), ~0.18 pA apart): a bracket then holds a
// handful of distinct values with ~10^6 samples each, far beyond any list.  The sample tells which values those are;
// the pass COUNTS the samples equal to one of them instead of copying them, and the finish treats each as one entry
// with a weight.  Exactness does not depend on the sample: a value it missed is simply copied as before.
#define N1H_SLOTS 256 // hash table of the sample's in-bracket keys (per minibatch; per block in LDS first)
#define N1H_MAX 32    // heavy keys handed to the pass
// per-minibatch words: [0, 256) key + 1, [256, 512) count, [512] number of heavy keys, [513] sample points inside the
// brackets, [514, 546) the keys (ascending),
// [546, 578) their counts over the whole minibatch (filled by k_n1_fused)
#define N1H_WORDS 640
#define N1H_NH 512
#define N1H_KEYS 514
#define N1H_CNTS 546

static __device__ __forceinline__ bool n1h_insert(uint32_t *tab, uint32_t key, uint32_t cnt)
{
    uint32_t slot = (key * 2654435761u) >> 24;
    for (int p = 0; p < 24; p++) {
        const uint32_t old = atomicCAS(&tab[slot], 0u, key + 1u);
        if (old == 0u || old == key + 1u) { atomicAdd(&tab[N1H_SLOTS + slot], cnt); return true; }
        slot = (slot + 1u) & (N1H_SLOTS - 1u);
    }
    return false;
}

// after the sample passes of a statistic: turn the key bracket into the float brackets of the fused pass
__global__ void k_n1_fuse_setup(MbState *__restrict__ mbs, N1Fused *__restrict__ fz, int n_mb, int mode)
{
    const int mb = blockIdx.x * blockDim.x + threadIdx.x;
    if (mb >= n_mb) return;
    MbState st = mbs[mb];
    N1Fused f = fz[mb];
    if (mode == 0) {
        f.ok = 0; f.med_s = 0.f; f.A0 = 0.f; f.A1 = 0.f; f.D0 = 0.f; f.D1 = 0.f; f.eps = 0.f; f.wide_ok = 0; f.D0w = 0.f; f.D1w = 0.f;
        if (st.status == ADP_MB_OK && st.ckw && st.cklo + st.ckw > st.cklo) {
            const float a_lo = key2f(st.cklo), a_hi = key2f(st.cklo + st.ckw);
            const float ms = key2f(st.cklo + (st.ckw >> 1));
            f.med_s = ms; f.A0 = a_lo - ms; f.A1 = a_hi - ms;
            f.eps = fmaxf(-f.A0, f.A1);
            f.ok = (f.A0 <= 0.f && f.A1 > 0.f && f.eps < __builtin_inff()) ? 1 : 0;
        }
        mbs[mb].med = f.med_s; // the MAD's sample passes measure |x - med_s|
    } else {
        if (f.ok) {
            if (st.ckw && st.cklo + st.ckw > st.cklo) {
                const float d_lo = key2f(st.cklo), d_hi = key2f(st.cklo + st.ckw);
                const float e = f.eps * 1.01f + 1e-6f * fabsf(f.med_s) + 1e-30f;
                f.D0 = d_lo - e; f.D1 = d_hi + e;
                f.ok = (f.D0 > f.eps && f.D1 > f.D0 && f.D1 < __builtin_inff()) ? 1 : 0; // the median bracket lies inside the inner zone
                // The zone the finish can use shrinks by |med - med_s| <= eps on either side.  With tied data the sample's
                // distances |x - med_s| themselves sit up to eps away from the true |x - med| (all of one level move
                // together), so the band needs twice the margin; it holds few distinct values then, and they are counted.
                const float e2 = f.eps * 2.02f + 1e-6f * fabsf(f.med_s) + 1e-30f;
                f.D0w = d_lo - e2; f.D1w = d_hi + e2;
                f.wide_ok = (f.ok && f.D0w > f.eps && f.D1w > f.D0w && f.D1w < __builtin_inff()) ? 1 : 0;
            } else f.ok = 0;
        }
        // leave the state as n1_select.h's own sample passes expect to find it
        mbs[mb].med = 0.f; mbs[mb].ckw = 0; mbs[mb].cklo = 0; mbs[mb].kbase = 0; mbs[mb].krem = 0; mbs[mb].bad = 0; mbs[mb].done = 0;
    }
    fz[mb] = f;
}

// The sample once more (same rows and columns as the sample passes): the keys of the points that the pass would copy,
// with their multiplicities -- per block in LDS, then only the repeated ones (>= 4 in this block's share) to the
// minibatch's table.
template <class SIG>
__global__ void __launch_bounds__(N1_THREADS) k_n1_heavy_scan(SIG sig, int n_reads, int m, int T, int mbsize,
                                                               const MbState *__restrict__ mbs, const N1Fused *__restrict__ fz,
                                                               uint32_t *__restrict__ hv, int row_step, int col_div, int pdiv,
                                                               const int32_t *__restrict__ full_len)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab_[2 * N1H_SLOTS];
    const int mb = blockIdx.y;
    if (mbs[mb].status != ADP_MB_OK) return;
    const N1Fused f = fz[mb];
    if (!f.ok) return;
    for (int i = threadIdx.x; i < 2 * N1H_SLOTS; i += N1_THREADS) tab_[i] = 0;
    __syncthreads();
    const int r0 = mb * mbsize;
    const int r1 = min(n_reads, r0 + mbsize);
    const float D0 = f.wide_ok ? f.D0w : f.D0, D1 = f.wide_ok ? f.D1w : f.D1; // (the band the pass will use if keys turn out heavy)
    uint32_t nin = 0;
    for (long long r = r0 + (long long)blockIdx.x * row_step; r < r1; r += (long long)gridDim.x * row_step) {
        const typename SIG::Row row = sig.row(r, m);
        // (the pieces of the sample passes, n1_select.h)
        const int npc = col_div > 1 ? col_div : 1;
        const int Rg = col_div > 1 ? ((T / col_div) & ~3) : 0, Tp = col_div > 1 ? ((T / (col_div * pdiv)) & ~3) : T;
        const int rot = col_div > 1 ? (int)(((r - r0) / row_step) % pdiv) : 0;
        auto visit = [&](float x) {
            const float u = x - f.med_s, t = fabsf(u);
            const bool inner = t < D0;
            const bool take = inner ? (u >= f.A0 && u < f.A1) : (t <= D1); // (as n1f_count)
            if (take) {
                nin++;
                const uint32_t key = f2key(x);
                uint32_t slot = (key * 2654435761u) >> 24;
                for (int p = 0; p < 6; p++) { // a full table (data without ties) just drops the point
                    const uint32_t old = atomicCAS(&tab_[slot], 0u, key + 1u);
                    if (old == 0u || old == key + 1u) { atomicAdd(&tab_[N1H_SLOTS + slot], 1u); break; }
                    slot = (slot + 1u) & (N1H_SLOTS - 1u);
                }
            }
        };
        const int tot = npc * Tp;
        int Te = T;
        if (full_len) { const int fl = full_len[r]; Te = fl < T ? (fl > 0 ? fl : 0) : T; }
        for (int i = threadIdx.x; i < tot; i += N1_THREADS) {
            const int pc = i / Tp, j = i - pc * Tp;
            if (pc * Rg + rot * Tp + j >= Te) continue;
            visit(row[(size_t)pc * Rg + (size_t)rot * Tp + j]);
        }
    }
    __syncthreads();
    uint32_t *g = hv + (size_t)mb * N1H_WORDS;
    nin = (uint32_t)wave_sum((int)nin);
    if (lane_id() == 0 && nin) atomicAdd(&g[N1H_NH + 1], nin);
    for (int sidx = threadIdx.x; sidx < N1H_SLOTS; sidx += N1_THREADS) {
        const uint32_t c = tab_[N1H_SLOTS + sidx];
        if (c >= 4u) n1h_insert(g, tab_[sidx] - 1u, c);
    }
}

// one wave per minibatch: the (at most N1H_MAX) keys that carry at least 1/128 of the repeated sample points, ascending
__global__ void __launch_bounds__(64) k_n1_heavy_pick(const MbState *__restrict__ mbs, N1Fused *__restrict__ fz, uint32_t *__restrict__ hv,
                                                       int n_mb)
{
    __shared__ __attribute__((aligned(16))) uint32_t keys_[N1H_MAX];
    const int mb = blockIdx.x;
    const int ln = threadIdx.x;
    uint32_t *g = hv + (size_t)mb * N1H_WORDS;
    uint32_t k[4], c[4];
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { k[j] = g[4 * ln + j]; c[j] = k[j] ? g[N1H_SLOTS + 4 * ln + j] : 0u; tot += c[j]; }
    tot = (uint32_t)wave_sum((int)tot);
    const uint32_t thr = tot / 128u > 16u ? tot / 128u : 16u;
    int nh = 0;
    uint32_t hsum = 0; // sample points on the chosen keys
    if (mbs[mb].status == ADP_MB_OK && fz[mb].ok) {
        for (; nh < N1H_MAX; nh++) {
            uint32_t bc = 0; int bj = -1;
#pragma unroll
            for (int j = 0; j < 4; j++) if (c[j] >= thr && c[j] > bc) { bc = c[j]; bj = j; }
            const uint32_t wc = wave_max(bc);
            if (wc == 0u) break;
            hsum += wc;
            const unsigned long long mk = __ballot(bc == wc);
            const int src = __ffsll((long long)mk) - 1;
            if (ln == src) {
                keys_[nh] = k[bj] - 1u;
#pragma unroll
                for (int j = 0; j < 4; j++) if (j == bj) c[j] = 0u;
            }
        }
    }
    __syncthreads();
    if (ln < nh) {
        const uint32_t mine = keys_[ln];
        int rank = 0;
        for (int j = 0; j < nh; j++) rank += keys_[j] < mine ? 1 : 0;
        g[N1H_KEYS + rank] = mine;
    }
    // Heavy keys are worth their lookups only where they carry the bulk of the samples to copy (quantised data); a few
    // repeated values among mostly distinct ones are left to the lists.
    if ((unsigned long long)hsum * 2ull < (unsigned long long)g[N1H_NH + 1]) nh = 0;
    __syncthreads();
    if (ln < N1H_MAX) g[N1H_CNTS + ln] = 0u;
    if (ln == 0) {
        g[N1H_NH] = (uint32_t)nh;
        // tied data -- nearly every sample point to copy sits on a heavy key: the wide band (its extra members are counted,
        // not copied); a few repeated values among otherwise distinct ones keep the narrow one
        if (nh > 0 && fz[mb].wide_ok && (unsigned long long)hsum * 10ull >= (unsigned long long)g[N1H_NH + 1] * 9ull) {
            fz[mb].D0 = fz[mb].D0w; fz[mb].D1 = fz[mb].D1w;
        }
    }
}

struct N1FAcc { uint32_t nvalid, nbelow, ninner; };

// per sample, free of branches: the three counts, and whether the sample has to be copied out (1.5 % of them)
static __device__ __forceinline__ uint32_t n1f_count(float x, float med_s, float A0, float A1, float D0, float D1, N1FAcc &a)
{
    const float u = x - med_s;
    const float t = fabsf(u);
    // pure mask arithmetic (no selects between conditions: the compiler turned `inner ? ... : ...` into exec-masked branches per
    // sample).  The median bracket lies inside the inner zone (k_n1_fuse_setup: eps < D0), so [A0, A1) needs no `inner` test.
    const bool valid = t == t;
    const bool ge0 = u >= A0, lt1 = u < A1, ged = t >= D0, led = t <= D1;
    a.nvalid += valid ? 1u : 0u;
    a.nbelow += (valid & !ge0) ? 1u : 0u;
    a.ninner += (valid & !ged) ? 1u : 0u;
    const bool take = (ge0 & lt1) | (ged & led); // (every comparison is false for a NaN)
    return take ? 1u : 0u;
}
// the copy itself: median bracket -> first list, MAD band -> second
// a heavy key is counted, not copied (out of line: the pass's inner loop keeps its registers)
static __device__ __noinline__ bool n1f_heavy(float x, int nh, const LDS uint32_t *hkeys, LDS uint32_t *hcnt)
{
    const uint32_t key = f2key(x);
    int lo = 0, hi = nh;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (hkeys[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo < nh && hkeys[lo] == key) { __hip_atomic_fetch_add(&hcnt[lo], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return true; }
    return false;
}
static __device__ __forceinline__ void n1f_copy(float x, float med_s, float D0, LDS float *cbm, LDS float *cbb, LDS uint32_t *cnt2,
                                                int nh, const LDS uint32_t *hkeys, LDS uint32_t *hcnt)
{
    if (nh && n1f_heavy(x, nh, hkeys, hcnt)) return;
    const bool inner = fabsf(x - med_s) < D0;
    uint32_t slot = __hip_atomic_fetch_add(cnt2 + (inner ? 0 : 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (inner) { if (slot < N1F_LDS_M) cbm[slot] = x; }
    else if (slot < N1F_LDS_B) cbb[slot] = x;
}
static __device__ __forceinline__ void n1f_account(float x, float med_s, float A0, float A1, float D0, float D1, N1FAcc &a,
                                                   LDS float *cbm, LDS float *cbb, LDS uint32_t *cnt2, int nh, const LDS uint32_t *hkeys,
                                                   LDS uint32_t *hcnt)
{
    if (n1f_count(x, med_s, A0, A1, D0, D1, a)) n1f_copy(x, med_s, D0, cbm, cbb, cnt2, nh, hkeys, hcnt);
}

// grid = (blocks_per_minibatch, n_minibatch); block = N1_THREADS
template <class SIG>
__global__ void __launch_bounds__(N1_THREADS) k_n1_fused(SIG sig, int n_reads, int m, int T, int mbsize,
                                                          const MbState *__restrict__ mbs, const N1Fused *__restrict__ fz,
                                                          unsigned long long *__restrict__ fcnt, float *__restrict__ cbuf,
                                                          uint32_t *__restrict__ hv, const int32_t *__restrict__ full_len)
{
    __shared__ __attribute__((aligned(16))) float cbm_[N1F_LDS_M];
    __shared__ __attribute__((aligned(16))) float cbb_[N1F_LDS_B];
    __shared__ __attribute__((aligned(16))) uint32_t cnt2_[2], base2[2];
    __shared__ __attribute__((aligned(16))) uint32_t hkeys_[N1H_MAX], hcnt_[N1H_MAX];
    const LDS uint32_t *hkeys = (const LDS uint32_t *)hkeys_;
    LDS uint32_t *hcnt = (LDS uint32_t *)hcnt_;
    LDS float *cbm = (LDS float *)cbm_;
    LDS float *cbb = (LDS float *)cbb_;
    LDS uint32_t *cnt2 = (LDS uint32_t *)cnt2_;
    const int mb = blockIdx.y;
    if (mbs[mb].status != ADP_MB_OK) return;
    const N1Fused f = fz[mb];
    if (!f.ok) return;
    if (threadIdx.x < 2) cnt2_[threadIdx.x] = 0;
    uint32_t *hg = hv + (size_t)mb * N1H_WORDS;
    const int nh = (int)hg[N1H_NH];
    if (threadIdx.x < N1H_MAX) { hkeys_[threadIdx.x] = threadIdx.x < nh ? hg[N1H_KEYS + threadIdx.x] : 0xffffffffu; hcnt_[threadIdx.x] = 0; }
    __syncthreads();
    const int r0 = mb * mbsize;
    const int r1 = min(n_reads, r0 + mbsize);
    const float med_s = f.med_s, A0 = f.A0, A1 = f.A1, D0 = f.D0, D1 = f.D1;
    N1FAcc a; a.nvalid = 0; a.nbelow = 0; a.ninner = 0;
    const bool vec = sig.vec_ok(m);
    float *dstm = cbuf + (size_t)mb * N1_CB_CAP;
    float *dstb = dstm + N1F_MCAP;
#define ACC(xx) n1f_account(xx, med_s, A0, A1, D0, D1, a, cbm, cbb, cnt2, nh, hkeys, hcnt)
    // hand the copied samples to the minibatch's lists: one global atomic per list, per block and per quarter row
    auto flush = [&]() {
        __syncthreads();
        const uint32_t cm = cnt2_[0], cbn = cnt2_[1];
        if (threadIdx.x == 0) {
            if (cm > N1F_LDS_M || cbn > N1F_LDS_B) { atomicAdd(&fcnt[N1F_NCNT * mb + 4], 1ull); base2[0] = 0xffffffffu; }
            else {
                base2[0] = cm ? (uint32_t)atomicAdd(&fcnt[N1F_NCNT * mb + 3], (unsigned long long)cm) : 0u;
                base2[1] = cbn ? (uint32_t)atomicAdd(&fcnt[N1F_NCNT * mb + 5], (unsigned long long)cbn) : 0u;
            }
        }
        __syncthreads();
        if (base2[0] != 0xffffffffu) {
            const uint32_t bm = base2[0], bb = base2[1];
            for (uint32_t i = threadIdx.x; i < cm; i += N1_THREADS) if (bm + i < N1F_MCAP) dstm[bm + i] = cbm[i];
            for (uint32_t i = threadIdx.x; i < cbn; i += N1_THREADS) if (bb + i < N1F_BCAP) dstb[bb + i] = cbb[i];
        }
        __syncthreads();
        if (threadIdx.x < 2) cnt2_[threadIdx.x] = 0;
        __syncthreads();
    };
    for (int r = r0 + blockIdx.x; r < r1; r += gridDim.x) {
        const typename SIG::Row row = sig.row(r, m);
        int Te = T; // ADP_TAILS_NAN: the NaN padding behind the read's end counts for nothing and is not read
        if (full_len) { const int fl = full_len[r]; Te = fl < T ? (fl > 0 ? fl : 0) : T; }
        if (vec) {
            const int T4 = Te >> 2;
            // the row in eight parts, the staging lists flushed after each: small lists leave LDS for 8 blocks per CU
            const int q4 = ((T4 + 7) / 8 + 2 * N1_THREADS - 1) / (2 * N1_THREADS) * (2 * N1_THREADS);
            for (int seg = 0; seg < T4; seg += q4) {
                const int send = min(T4, seg + q4);
                int i = seg + threadIdx.x;
                for (; i + 3 * N1_THREADS < send; i += 4 * N1_THREADS) { // four loads in flight per lane (-3 % against two)
                    float4 v0 = row.f4s_in(i), v1 = row.f4s_in(i + N1_THREADS), v2 = row.f4s_in(i + 2 * N1_THREADS), v3 = row.f4s_in(i + 3 * N1_THREADS);
                    const float e[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
                    uint32_t fl = 0;
#pragma unroll
                    for (int q = 0; q < 16; q++) fl |= n1f_count(e[q], med_s, A0, A1, D0, D1, a) << q;
                    while (fl) {
                        const int q = __ffs(fl) - 1;
                        fl &= fl - 1;
                        float x = e[0];
#pragma unroll
                        for (int z = 1; z < 16; z++) x = (q == z) ? e[z] : x;
                        n1f_copy(x, med_s, D0, cbm, cbb, cnt2, nh, hkeys, hcnt);
                    }
                }
                for (; i + N1_THREADS < send; i += 2 * N1_THREADS) { // two loads in flight per lane
                    float4 v = row.f4s_in(i), w = row.f4s_in(i + N1_THREADS);
                    const float e[8] = {v.x, v.y, v.z, v.w, w.x, w.y, w.z, w.w};
                    uint32_t fl = 0;
#pragma unroll
                    for (int q = 0; q < 8; q++) fl |= n1f_count(e[q], med_s, A0, A1, D0, D1, a) << q;
                    while (fl) { // the few samples to copy: one at a time, picked out of the eight registers
                        const int q = __ffs(fl) - 1;
                        fl &= fl - 1;
                        float x = e[0];
#pragma unroll
                        for (int z = 1; z < 8; z++) x = (q == z) ? e[z] : x;
                        n1f_copy(x, med_s, D0, cbm, cbb, cnt2, nh, hkeys, hcnt);
                    }
                }
                for (; i < send; i += N1_THREADS) { float4 v = row.f4_in(i); ACC(v.x); ACC(v.y); ACC(v.z); ACC(v.w); }
                if (send == T4) for (int j = (T4 << 2) + threadIdx.x; j < Te; j += N1_THREADS) ACC(row[j]);
                flush();
            }
            if (T4 == 0 && Te > 0) { for (int j = threadIdx.x; j < Te; j += N1_THREADS) ACC(row[j]); flush(); }
        } else {
            for (int j = threadIdx.x; j < Te; j += N1_THREADS) ACC(row[j]);
            flush();
        }
    }
#undef ACC
    unsigned long long nv = (unsigned long long)wave_sum((int)a.nvalid), nb = (unsigned long long)wave_sum((int)a.nbelow),
                       ni = (unsigned long long)wave_sum((int)a.ninner);
    if (lane_id() == 0) {
        if (nv) atomicAdd(&fcnt[N1F_NCNT * mb], nv);
        if (nb) atomicAdd(&fcnt[N1F_NCNT * mb + 1], nb);
        if (ni) atomicAdd(&fcnt[N1F_NCNT * mb + 2], ni);
    }
    __syncthreads();
    if ((int)threadIdx.x < nh && hcnt_[threadIdx.x]) atomicAdd(&hg[N1H_CNTS + threadIdx.x], hcnt_[threadIdx.x]);
}

// ---- finish: counting selection in LDS -------------------------------------------------------------------------
struct N1Sel {
    uint32_t wtot[16];
    uint32_t s_before, s_total, s_below, s_clo, s_chi;
    int s_bin, s_low;
};

// block-wide (1024 threads): total of hist[0..nb) and the bin holding rank k (-1: k >= total)
static __device__ void n1f_find(LDS uint32_t *hist, int nb, uint32_t k, LDS N1Sel *S)
{
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024, b0 = tid * per;
    uint32_t s = 0;
    for (int j = 0; j < per; j++) if (b0 + j < nb) s += hist[b0 + j];
    const uint32_t incl = (uint32_t)wave_scan_incl((int)s);
    if (lane_id() == 63) S->wtot[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int w = 0; w < 16; w++) { uint32_t t = S->wtot[w]; S->wtot[w] = run; run += t; }
        S->s_total = run; S->s_bin = -1; S->s_before = 0; S->s_low = -1;
    }
    __syncthreads();
    const uint32_t excl = S->wtot[tid >> 6] + incl - s;
    if (s && k >= excl && k < excl + s) {
        uint32_t cum = excl;
        for (int j = 0; j < per; j++) {
            uint32_t c = hist[b0 + j];
            if (k < cum + c) { S->s_bin = b0 + j; S->s_before = cum; break; }
            cum += c;
        }
    }
    __syncthreads();
}

// Block-wide exact selection among xs[0..n) by the keys keyf(x), restricted to the zone [klo, khi] (inclusive):
// *c_lo / *c_hi = items with a key below / above the zone, *n_zone = items inside; if 0 <= kz(c_lo) < n_zone (kz =
// the wanted rank inside the zone, given c_lo) the key of that rank and of the rank before it (needs kz >= 1) are
// returned.  Counting sort on a key histogram in LDS, 15 bits of the zone's span per level.
template <class KF, class RK>
static __device__ bool n1f_select(LDS uint32_t *hist, LDS N1Sel *S, const float *__restrict__ xs, uint32_t n, KF keyf, uint32_t klo,
                                  uint32_t khi, RK rank_in_zone, bool need_prev, uint32_t *out_k, uint32_t *out_km1,
                                  uint32_t *c_lo_out, uint32_t *c_hi_out, int nh = 0, const LDS float *hval = nullptr,
                                  const LDS uint32_t *hwgt = nullptr)
{
    // (hval[j], hwgt[j]), j < nh: the heavy values of this list -- one entry each, counted hwgt[j] times (0: not in this list)
    const int tid = threadIdx.x;
    const unsigned long long span = (unsigned long long)khi - klo + 1ull;
    int bits = 0; while ((1ull << bits) < span) bits++;
    int sh = bits > 15 ? bits - 15 : 0;
    uint32_t base = klo;
    unsigned long long width = span; // keys still in play: [base, base + width)
    long long kz = 0;
    bool first = true;
    const float4 *xs4 = reinterpret_cast<const float4 *>(xs);
    for (;;) {
        const int nb = (int)((width + ((1ull << sh) - 1ull)) >> sh);
        for (int i = tid; i < nb; i += 1024) hist[i] = 0;
        if (tid == 0) { S->s_below = 0; if (first) { S->s_clo = 0; S->s_chi = 0; } }
        __syncthreads();
        uint32_t clo = 0, chi = 0, below = 0;
        auto visit_w = [&](float x, uint32_t w) {
            const uint32_t key = keyf(x);
            if (key < base) {
                if (key < klo) clo += w;
                else if (key + 1u > below) below = key + 1u; // inside the zone, below the range in play (+1: 0 = none)
            } else {
                const unsigned long long d = (unsigned long long)key - base;
                if (d < width) __hip_atomic_fetch_add(&hist[(uint32_t)(d >> sh)], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (key > khi) chi += w;
            }
        };
        auto visit = [&](float x) { visit_w(x, 1u); };
        const uint32_t n4 = n >> 2;
        uint32_t i = tid;
        for (; i + 7 * 1024 < n4; i += 8 * 1024) { // eight 16-byte loads in flight per thread: one block per minibatch
            float4 v[8];                             // has to pull its copied samples through a single CU
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = xs4[i + u * 1024];
#pragma unroll
            for (int u = 0; u < 8; u++) { visit(v[u].x); visit(v[u].y); visit(v[u].z); visit(v[u].w); }
        }
        for (; i + 1024 < n4; i += 2048) {
            float4 v = xs4[i], w = xs4[i + 1024];
            visit(v.x); visit(v.y); visit(v.z); visit(v.w); visit(w.x); visit(w.y); visit(w.z); visit(w.w);
        }
        for (; i < n4; i += 1024) { float4 v = xs4[i]; visit(v.x); visit(v.y); visit(v.z); visit(v.w); }
        for (uint32_t j = (n4 << 2) + tid; j < n; j += 1024) visit(xs[j]);
        for (int j = tid; j < nh; j += 1024) if (hwgt[j]) visit_w(hval[j], hwgt[j]);
        below = wave_max(below);
        if (lane_id() == 0 && below) atomicMax((uint32_t *)&S->s_below, below);
        if (first) {
            clo = (uint32_t)wave_sum((int)clo); chi = (uint32_t)wave_sum((int)chi);
            if (lane_id() == 0) { if (clo) atomicAdd((uint32_t *)&S->s_clo, clo); if (chi) atomicAdd((uint32_t *)&S->s_chi, chi); }
        }
        __syncthreads();
        if (first) {
            *c_lo_out = S->s_clo; *c_hi_out = S->s_chi;
            kz = rank_in_zone(S->s_clo);
            if (kz < (need_prev ? 1 : 0)) return false;
        }
        n1f_find(hist, nb, (uint32_t)kz, S);
        if (S->s_bin < 0) return false; // (first level: kz >= n_zone)
        const int bin = S->s_bin;
        const uint32_t before = S->s_before;
        if (sh == 0) {
            // largest non-empty bin below `bin`
            { const int per = (nb + 1023) / 1024, b0 = tid * per; int low = -1;
              for (int j = 0; j < per; j++) { int b = b0 + j; if (b < bin && b < nb && hist[b]) low = b; }
              if (low >= 0) atomicMax((int *)&S->s_low, low); }
            __syncthreads();
            const uint32_t k1 = base + (uint32_t)bin;
            uint32_t k0 = k1;
            if ((uint32_t)kz == before) { // first of its key: the rank before it is the largest key below
                if (S->s_low >= 0) k0 = base + (uint32_t)S->s_low;
                else if (S->s_below) k0 = S->s_below - 1u;
                else if (need_prev) return false; // (cannot happen with kz >= 1)
            }
            *out_k = k1; *out_km1 = k0;
            __syncthreads();
            return true;
        }
        base += (uint32_t)bin << sh;
        width = 1ull << sh;
        kz -= before;
        sh = sh > 15 ? sh - 15 : 0;
        first = false;
        __syncthreads();
    }
}

// one block (1024 threads) per minibatch
__global__ void __launch_bounds__(1024) k_n1_fused_finish(MbState *__restrict__ mbs, const N1Fused *__restrict__ fz,
                                                           unsigned long long *__restrict__ fcnt, const float *__restrict__ cbuf,
                                                           double thresh, const uint32_t *__restrict__ hv)
{
    __shared__ __attribute__((aligned(16))) uint32_t hist_[N1F_BINS];
    __shared__ __attribute__((aligned(16))) N1Sel S_;
    __shared__ __attribute__((aligned(16))) float hval_[N1H_MAX];
    __shared__ __attribute__((aligned(16))) uint32_t hwm_[N1H_MAX], hwb_[N1H_MAX], hsum_[2];
    LDS uint32_t *hist = (LDS uint32_t *)hist_;
    LDS N1Sel *S = (LDS N1Sel *)&S_;
    const int mb = blockIdx.x;
    const int tid = threadIdx.x;
    const MbState st = mbs[mb];
    const N1Fused f = fz[mb];
    const unsigned long long n_valid = fcnt[N1F_NCNT * mb], n_below = fcnt[N1F_NCNT * mb + 1], n_inner = fcnt[N1F_NCNT * mb + 2],
                             ovf = fcnt[N1F_NCNT * mb + 4];
    unsigned long long n_cm = fcnt[N1F_NCNT * mb + 3], n_cb = fcnt[N1F_NCNT * mb + 5];
    // heavy values: counted by the pass, each one entry of its list (median bracket: |x - med_s| < D0, else MAD band)
    const uint32_t *hg = hv + (size_t)mb * N1H_WORDS;
    const int nh = (st.status == ADP_MB_OK && f.ok) ? (int)hg[N1H_NH] : 0;
    if (tid < 2) hsum_[tid] = 0;
    __syncthreads();
    if (tid < N1H_MAX) {
        float x = 0.f; uint32_t wm = 0, wb = 0;
        if (tid < nh) {
            x = key2f(hg[N1H_KEYS + tid]);
            const uint32_t c = hg[N1H_CNTS + tid];
            if (fabsf(x - f.med_s) < f.D0) wm = c; else wb = c;
            if (wm) atomicAdd(&hsum_[0], wm);
            if (wb) atomicAdd(&hsum_[1], wb);
        }
        hval_[tid] = x; hwm_[tid] = wm; hwb_[tid] = wb;
    }
    __syncthreads();
    if (tid < N1F_NCNT) fcnt[N1F_NCNT * mb + tid] = 0; // ready for the next call
    if (st.status != ADP_MB_OK || !f.ok) return;
#ifdef N1F_DEBUG_PRINT
    if (tid == 0 && mb < 6) printf("mb %d nvalid %llu below %llu inner %llu n_cm %llu n_cb %llu ovf %llu med_s %.6f A0 %.6f A1 %.6f D0 %.6f D1 %.6f eps %.6f nh %d\n", mb, n_valid, n_below, n_inner, n_cm, n_cb, ovf, f.med_s, f.A0, f.A1, f.D0, f.D1, f.eps, nh);
#endif
    if (tid == 0) { atomicAdd(&g_dbg[5], 1ull); atomicAdd(&g_dbg[22], (unsigned long long)nh); atomicAdd(&g_dbg[23], (unsigned long long)hsum_[0] + hsum_[1]);
                    if (ovf) atomicAdd(&g_dbg[19], 1ull); if (n_cm > N1F_MCAP) atomicAdd(&g_dbg[18], 1ull); if (n_cb > N1F_BCAP) atomicAdd(&g_dbg[17], 1ull); }
    if (ovf || n_cm > N1F_MCAP || n_cb > N1F_BCAP || n_valid < 4) { if (tid == 0) atomicAdd(&g_dbg[6], 1ull); return; }
    const float *xm = cbuf + (size_t)mb * N1_CB_CAP;
    const float *xb = xm + N1F_MCAP;
    const uint32_t l_cm = (uint32_t)n_cm, l_cb = (uint32_t)n_cb; // list lengths
    n_cm += hsum_[0]; n_cb += hsum_[1];                           // members of the brackets, heavy ones included
    const LDS float *hval = (const LDS float *)hval_;
    const LDS uint32_t *hwm = (const LDS uint32_t *)hwm_, *hwb = (const LDS uint32_t *)hwb_;
    const float med_s = f.med_s, D0 = f.D0, D1 = f.D1;
    const bool even = (n_valid & 1ull) == 0;
    const unsigned long long k = n_valid / 2;

    // ---- median: rank k - below among the copied samples of the first bracket (a little slack on the key range:
    // the bracket is defined on u = x - med_s; anything outside the range makes the attempt fail, never wrong)
    uint32_t kk = 0, kkm1 = 0, c_lo = 0, c_hi = 0;
    {
        const uint32_t ka = f2key(med_s + f.A0), kb = f2key(med_s + f.A1);
        const uint32_t klo = ka > 64u ? ka - 64u : 0u, khi = kb < 0xffffffffu - 64u ? kb + 64u : 0xffffffffu;
        const bool okm = n1f_select(hist, S, xm, l_cm, [](float x) { return f2key(x); }, klo, khi,
                                    [&](uint32_t) { return (k >= n_below && k - n_below < n_cm) ? (long long)(k - n_below) : -1ll; },
                                    even, &kk, &kkm1, &c_lo, &c_hi, nh, hval, hwm);
        if (!okm || c_lo || c_hi) { if (tid == 0) atomicAdd(&g_dbg[6], 1ull); return; }
    }
    float med = key2f(kk);
    if (even) med = (key2f(kkm1) + med) / 2.0f;

    // ---- MAD: zone (vlo, vhi) of v = |x - med| whose members were all copied
    const double delta = fabs((double)med - (double)med_s);
    const double tlo = ((double)D0 + delta) * (1.0 + 1e-12), thi = ((double)D1 - delta) * (1.0 - 1e-12);
    float vlo = (float)tlo; if ((double)vlo < tlo) vlo = __uint_as_float(__float_as_uint(vlo) + 1u); // round up   (vlo > 0)
    float vhi = (float)thi; if ((double)vhi > thi) vhi = __uint_as_float(__float_as_uint(vhi) - 1u); // round down (vhi > 0)
    if (!(thi > 0.0) || !(vhi > vlo) || f2key(vhi) - f2key(vlo) < 2u) { if (tid == 0) atomicAdd(&g_dbg[7], 1ull); return; }
    {
        // samples with v <= vlo: every inner one plus the copied ones below the zone
        const bool okd = n1f_select(hist, S, xb, l_cb, [&](float x) { return f2key(fabsf(x - med)); }, f2key(vlo) + 1u,
                                    f2key(vhi) - 1u,
                                    [&](uint32_t clo) { unsigned long long cle = n_inner + clo; return k >= cle ? (long long)(k - cle) : -1ll; },
                                    even, &kk, &kkm1, &c_lo, &c_hi, nh, hval, hwb);
        if (!okd) { if (tid == 0) atomicAdd(&g_dbg[7], 1ull); return; }
    }
    float mad = key2f(kk);
    if (even) mad = (key2f(kkm1) + mad) / 2.0f;
    if (tid == 0) {
        mbs[mb].n_valid = n_valid;
        mbs[mb].med = med;
        mbs[mb].mad = mad;
        const double dmed = (double)med, dmad = (double)mad;
        mbs[mb].lo = (float)(dmed - dmad * thresh);
        mbs[mb].hi = (float)(dmed + dmad * thresh);
        if (mad == 0.0f) mbs[mb].status = ADP_MB_MAD_ZERO;
        mbs[mb].fused = 1;
    }
}
