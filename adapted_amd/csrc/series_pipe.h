// series_pipe.h -- V4's moving-window series (bottleneck.move_var / move_mean, reference adapted/detect/mvs.py:93-106) for LONG
// slices as a PIPELINE of waves (round 4; k_mvs_series_wave, validate.h, is the one-wave-per-recurrence form and stays as the
// fallback for windows this one does not take).
//
// The recurrences are strictly sequential float32 chains per read (same operations, same order as bn_move_var / bn_move_mean):
//     delta = a_i - a_old;  a_old -= mean;  mean += delta * (1/w);  a_i -= mean;  ssq += (a_i + a_old) * delta;  ssq = ssq < 0 ? 0 : ssq
// so a launch cannot end before its longest slice has been walked, one step after the other: 195 000 steps at the 200 k window.
// k_mvs_series_wave spends ~88 cycles per step on them -- ONE wave issues all eleven instructions of a step (a wave64 instruction
// takes 4 cycles to issue, a dependent one ~8), its LDS ring copies and its loads and stores.  But only two short chains are truly
// serial: the mean (one add per step) and the clamped sum of squares (add, compare, select).  Here they run in DIFFERENT waves of a
// workgroup, on different SIMDs, each with nothing else to issue:
//     wave 0  loader      coalesced 16-byte loads of the next chunk (4 reads per instruction) -> the reads' LDS rings
//     wave 1  mean        move_mean: asum += a_i - a_old                                    -> outm
//     wave 2  var, part 1 delta, the mean chain, t = (a_i' + a_old') * delta                -> tbuf     (7 instructions per step)
//     wave 3  var, part 2 ssq = max0(ssq + t), out = ssq * (1/w), one chunk behind part 1   -> outv     (4 instructions per step)
//     wave 4  storer      outv of the chunk before -> global memory, 16-byte stores (4 reads per instruction)
//     wave 5  storer      outm likewise
// (waves 0 / 4 and 1 / 5 share a SIMD, the two halves of the variance have one each); one barrier per chunk of 64 steps, every
// buffer between two waves double.  Measured (24 000 reads at the 200 k window, longest slice 195 000 steps = 3050 chunks): 12.1 ms against
// 12.8 for k_mvs_series_wave on the same ordered reads; with parts of it left out (results wrong, timing only): no mean wave 10.2, no
// part 1 9.0, no part 2 10.5, no storers 10.0, neither part 7.9, no arithmetic at all 6.4 ms -- the skeleton (loader, storers, one barrier
// per chunk) costs 2 us per chunk, and the roles' costs ADD although they sit on different SIMDs: at this occupancy (six waves on a CU,
// most CUs' SIMDs idle) an instruction costs ~3.3 ns to issue and ~10 ns when it depends on its predecessor (tools/wave_simd_placement.hip),
// and every wave has a few hundred of them per chunk behind the same barrier.  A lane is a read (SP_G = 48 of the 64: LDS, 128 KB per workgroup), the reads come ordered by
// falling length (k_series_plan / k_series_order), so a wave's chains end together and the longest start first.
#pragma once
#include "validate.h"

#define SP_G 48            // reads per workgroup
#define SP_CH 64           // steps per chunk
#define SP_RB 256          // ring: samples of a read kept in LDS (>= longest window + 2 chunks)
#define SP_S (SP_RB + 4)   // ring row pitch in floats (4 x odd: the lanes' 16-byte accesses fall into different bank groups)
#define SP_SO (SP_CH + 4)  // pitch of the chunk buffers (t, outv, outm)
#define SP_MAXW (SP_RB - 2 * SP_CH)
#define SP_THREADS 384
#define SP_LDS_FLOATS (SP_G * SP_S + 6 * SP_G * SP_SO + SP_G)

static __host__ __device__ inline bool sp_takes(int wv, int wm)
{
    return wv >= 4 && wm >= 4 && (wv & 3) == 0 && (wm & 3) == 0 && wv <= SP_MAXW && wm <= SP_MAXW;
}

typedef float sp_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float sp_f2 __attribute__((ext_vector_type(2)));

// The barrier between the waves of the pipeline: LDS traffic done (lgkmcnt), then s_barrier.  NOT __syncthreads(): that also waits
// for the wave's outstanding global loads and stores (vmcnt(0)) -- the loader's prefetch and the storers' writes would be waited for
// in every chunk, a memory round trip (~2-4 us) per 64 steps (measured: 12.0 ms per launch with it, the chains idle 2/3 of the time)
#ifdef ADP_PHASE_TIMING
// (diagnostic build: per role of workgroup 0 -- the longest chains -- the cycles between leaving a barrier and reaching the next one,
// g_dbg[24 + 2 wave], and the cycles of the whole loop, g_dbg[25 + 2 wave]; tools/experiments/series_phase_shares.py)
static __device__ inline long long sp_now() { long long t; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
#define SP_BARRIER() do { const long long tb_ = sp_now(); sp_work_ += tb_ - sp_t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); sp_t_ = sp_now(); } while (0)
#else
#define SP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// grid = ceil(n_reads / SP_G); block = SP_THREADS; dynamic LDS = SP_LDS_FLOATS floats.  a_plan / n_plan / perm: k_series_plan, k_series_order
__global__ void __launch_bounds__(SP_THREADS) k_mvs_series_pipe(const float *__restrict__ sigs, int n_reads, int m, const int32_t *__restrict__ a_plan,
                                                                const int32_t *__restrict__ n_plan, const int32_t *__restrict__ perm, int wv, int wm,
                                                                float *__restrict__ series, int cap, int8_t *__restrict__ have)
{
    extern __shared__ __attribute__((aligned(16))) float sp_raw[]; // (aligned: behind 580 bytes of static LDS every 16-byte access of the rings was misaligned -- ~270 cycles each, round 5)
    __shared__ __attribute__((aligned(16))) int32_t rid_[SP_G], a_[SP_G], n_[SP_G];
    __shared__ int nmax_;
    LDS float *ring = (LDS float *)sp_raw;
    LDS float *tbuf = ring + SP_G * SP_S;
    LDS float *outv = tbuf + 2 * SP_G * SP_SO;
    LDS float *outm = outv + 2 * SP_G * SP_SO;
    LDS float *s0 = outm + 2 * SP_G * SP_SO;
    LDS int32_t *rid_of = (LDS int32_t *)rid_, *a_of = (LDS int32_t *)a_, *n_of = (LDS int32_t *)n_;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln = threadIdx.x & 63;
    const int r0 = blockIdx.x * SP_G;
    if (threadIdx.x == 0) nmax_ = 0;
    __syncthreads();
    if (threadIdx.x < SP_G) {
        const int idx = r0 + threadIdx.x;
        const int rd = perm[idx < n_reads ? idx : r0];
        const int nn = idx < n_reads ? n_plan[rd] : 0;
        rid_of[threadIdx.x] = rd;
        a_of[threadIdx.x] = idx < n_reads ? a_plan[rd] : 0;
        n_of[threadIdx.x] = nn;
        __hip_atomic_fetch_max((LDS int *)&nmax_, nn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    const int nmax = nmax_;
    if (nmax <= 0) return; // (uniform: nothing in this group has series)
    const int nchunks = (nmax + SP_CH - 1) / SP_CH;
    constexpr int MASK = SP_RB - 1, NU = SP_G / 4;
#ifdef ADP_PHASE_TIMING
    long long sp_work_ = 0, sp_t_ = sp_now();
    const long long sp_t0_ = sp_t_;
    struct SpStamp { long long &w, &t; const long long &t0; int wave, ln; __device__ ~SpStamp() { if (blockIdx.x == 0 && ln == 0) { g_dbg[48 + 2 * wave] = (unsigned long long)w; g_dbg[49 + 2 * wave] = (unsigned long long)(sp_now() - t0); } } } sp_stamp_{sp_work_, sp_t_, sp_t0_, wave, ln};
    if (blockIdx.x == 0 && ln == 0) { // where the waves sit: HW_ID's simd_id [5:4] and cu_id [11:8], a nibble pair per wave
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        reinterpret_cast<unsigned char *>(&g_dbg[60])[wave] = (unsigned char)(((id >> 4) & 3) | (((id >> 8) & 15) << 4));
    }
#endif

    if (wave == 0) {
        // ---- loader: instruction u of a chunk brings 64 samples of the reads 4u .. 4u + 3, 16 lanes x 16 bytes each
        const int sub = ln & 15;
        const GLB float *xb[NU];
        int last[NU], nend[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int q = 4 * u + (ln >> 4);
            xb[u] = (const GLB float *)sigs + (size_t)rid_of[q] * m + a_of[q];
            const int nq = n_of[q];
            last[u] = nq >= 4 ? nq - 4 : 0; // (what lies behind a slice's end is never used: any readable address will do)
            nend[u] = nq;
        }
        // a chunk's loads are issued a WHOLE iteration before their samples go to LDS (two register sets in turn): issued at the end
        // of one iteration and consumed at the start of the next, every chunk waited a memory round trip (~2 us against ~0.8 us of chains)
        sp_f4u preA[NU], preB[NU];
        auto fetch = [&](int c, sp_f4u (&pre)[NU]) {
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int i = c * SP_CH + 4 * sub;
#if defined(SP_ABL) && (SP_ABL & 1) // (timing only: no global loads)
                pre[u] = (sp_f4u){(float)i, 1.f, 2.f, 3.f};
#else
                pre[u] = *reinterpret_cast<const GLB sp_f4u *>(xb[u] + (i < last[u] ? i : last[u]));
#endif
            }
        };
        auto put = [&](int c, const sp_f4u (&pre)[NU]) {
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int q = 4 * u + (ln >> 4), i = c * SP_CH + 4 * sub;
                // (a clamped load carries other samples than i .. i + 3: cells no chain reads, since i + 3 >= n there -- except
                // the vector that straddles the end: its leading samples are rebuilt below)
                LDS float *dst = ring + q * SP_S + (i & MASK);
                const int lim = last[u];
                if (i <= lim) *reinterpret_cast<LDS adp_v4f *>(dst) = (adp_v4f){pre[u].x, pre[u].y, pre[u].z, pre[u].w};
                else if (i < nend[u]) { // (a vector wholly behind the end: nothing reads its cells -- the lanes of ended reads pass)
                    // the load was taken at `lim` instead of i: sample i + j sits at position i + j - lim of the vector, if inside it
                    const int sh = i - lim;
                    const float v[4] = {pre[u].x, pre[u].y, pre[u].z, pre[u].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) { const int s = sh + j; float val = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; e++) if (e == s) val = v[e];
                        dst[j] = val; }
                }
            }
        };
        fetch(0, preA);
        fetch(1, preB);
        put(0, preA);
        fetch(2, preA);
        SP_BARRIER();                                   // (barrier P: chunk 0 lies in the rings)
        // iteration k: chunk k + 1 (loaded two fetches ago) goes to the rings, chunk k + 3 is requested
        for (int k = 0; k < nchunks + 2; k += 2) {
            put(k + 1, preB);
            fetch(k + 3, preB);
            SP_BARRIER();
            if (k + 1 < nchunks + 2) {
                put(k + 2, preA);
                fetch(k + 4, preA);
                SP_BARRIER();
            }
        }
    } else if (wave == 1) {
        // ---- move_mean (the chain: one add per step)
        const int g = ln < SP_G ? ln : 0;
        const int n = ln < SP_G ? n_of[g] : 0;
        const LDS float *mybuf = ring + g * SP_S;
        const float inv = (float)(1.0 / (double)wm);
        float asum = 0.f;
        SP_BARRIER();                                   // P
        for (int k = 0; k < nchunks + 2; k++) {
            const int i0 = k * SP_CH;
            LDS float *myout = outm + (k & 1) * SP_G * SP_SO + g * SP_SO;
            if (i0 < n) {
                const int hi = n < i0 + SP_CH ? n : i0 + SP_CH;
                int i = i0;
                for (; i < hi && i < wm; i++) { asum += mybuf[i & MASK]; if (i == wm - 1) myout[i - i0] = asum / (float)wm; }
                if (i == i0 && hi == i0 + SP_CH) {
                    // a whole chunk (nearly all of them): its 2 x 64 samples come to registers in one burst of LDS reads, the chain
                    // runs without a wait, the 64 results leave in one burst (with a group's reads one group ahead -- round 4 -- every
                    // group of four steps still waited ~0.2 us for them: 3.6 us per chunk where the instructions take 0.4)
                    adp_v4f a4[SP_CH / 4], o4[SP_CH / 4];
#pragma unroll
                    for (int j = 0; j < SP_CH / 4; j++) {
                        a4[j] = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[(i0 & MASK) + 4 * j]); // (a chunk never wraps in the ring)
                        o4[j] = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[(i0 + 4 * j - wm) & MASK]);
                    }
                    // (what is not on the chain goes two steps per instruction -- v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations)
#pragma unroll
                    for (int j = 0; j < SP_CH / 4; j++) {
                        const sp_f2 d01 = (sp_f2){a4[j].x, a4[j].y} - (sp_f2){o4[j].x, o4[j].y}, d23 = (sp_f2){a4[j].z, a4[j].w} - (sp_f2){o4[j].z, o4[j].w};
                        sp_f2 s01, s23;
                        asum += d01.x; s01.x = asum;
                        asum += d01.y; s01.y = asum;
                        asum += d23.x; s23.x = asum;
                        asum += d23.y; s23.y = asum;
                        const sp_f2 r01 = s01 * (sp_f2){inv, inv}, r23 = s23 * (sp_f2){inv, inv};
                        *reinterpret_cast<LDS adp_v4f *>(&myout[4 * j]) = (adp_v4f){r01.x, r01.y, r23.x, r23.y};
                    }
                } else {
                    for (; i + 4 <= hi; i += 4) {
                        const adp_v4f a4 = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[i & MASK]);
                        const adp_v4f o4 = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[(i - wm) & MASK]);
                        adp_v4f r4;
                        asum += a4.x - o4.x; r4.x = asum * inv;
                        asum += a4.y - o4.y; r4.y = asum * inv;
                        asum += a4.z - o4.z; r4.z = asum * inv;
                        asum += a4.w - o4.w; r4.w = asum * inv;
                        *reinterpret_cast<LDS adp_v4f *>(&myout[i - i0]) = r4;
                    }
                    for (; i < hi; i++) { asum += mybuf[i & MASK] - mybuf[(i - wm) & MASK]; myout[i - i0] = asum * inv; }
                }
            }
            SP_BARRIER();
        }
        if (ln < SP_G && n > 0 && asum != asum) have[rid_of[g]] = 0; // a NaN stays in the chain to its end: the series are withdrawn
    } else if (wave == 2) {
        // ---- move_var, part 1: the window fills (Welford), then per step delta, the mean chain and t = (a_i' + a_old') * delta
        const int g = ln < SP_G ? ln : 0;
        const int n = ln < SP_G ? n_of[g] : 0;
        const LDS float *mybuf = ring + g * SP_S;
        const float inv = (float)(1.0 / (double)wv);
        float amean = 0.f, assqdm = 0.f;
        int count = 0;
        SP_BARRIER();                                   // P
        for (int k = 0; k < nchunks + 2; k++) {
            const int i0 = k * SP_CH;
            LDS float *myt = tbuf + (k & 1) * SP_G * SP_SO + g * SP_SO;
            if (i0 < n) {
                const int hi = n < i0 + SP_CH ? n : i0 + SP_CH;
                int i = i0;
                for (; i < hi && i < wv; i++) {
                    const float ai = mybuf[i & MASK];
                    count++;
                    const float delta = ai - amean;
                    amean += delta / (float)count;
                    assqdm += delta * (ai - amean);
                    if (i == wv - 1) {
                        if (assqdm < 0) assqdm = 0;
                        // (the first output goes straight to part 2's buffer of this chunk: no storer reads that half yet -- the
                        // window ends inside chunk 0 or 1 -- and part 2 writes it from i = wv on only)
                        outv[(k & 1) * SP_G * SP_SO + g * SP_SO + (i - i0)] = assqdm / (float)count;
                        s0[g] = assqdm;
                    }
                }
                if (i == i0 && hi == i0 + SP_CH) {
                    // (a whole chunk from registers, as in the mean wave)
                    adp_v4f a4[SP_CH / 4], o4[SP_CH / 4];
#pragma unroll
                    for (int j = 0; j < SP_CH / 4; j++) {
                        a4[j] = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[(i0 & MASK) + 4 * j]);
                        o4[j] = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[(i0 + 4 * j - wv) & MASK]);
                    }
                    // per step: p = o - mean, mean += d / w (the chain), q = a - mean; d, d / w, q + p and (q + p) d go two steps per instruction
#pragma unroll
                    for (int j = 0; j < SP_CH / 4; j++) {
                        const sp_f2 a01 = {a4[j].x, a4[j].y}, a23 = {a4[j].z, a4[j].w}, o01 = {o4[j].x, o4[j].y}, o23 = {o4[j].z, o4[j].w};
                        const sp_f2 d01 = a01 - o01, d23 = a23 - o23;
                        const sp_f2 e01 = d01 * (sp_f2){inv, inv}, e23 = d23 * (sp_f2){inv, inv};
                        sp_f2 p01, q01, p23, q23;
                        p01.x = o01.x - amean; amean += e01.x; q01.x = a01.x - amean;
                        p01.y = o01.y - amean; amean += e01.y; q01.y = a01.y - amean;
                        p23.x = o23.x - amean; amean += e23.x; q23.x = a23.x - amean;
                        p23.y = o23.y - amean; amean += e23.y; q23.y = a23.y - amean;
                        const sp_f2 t01 = (q01 + p01) * d01, t23 = (q23 + p23) * d23;
                        *reinterpret_cast<LDS adp_v4f *>(&myt[4 * j]) = (adp_v4f){t01.x, t01.y, t23.x, t23.y};
                    }
                    i = hi;
                }
                for (; i + 4 <= hi; i += 4) {
                    const adp_v4f a4 = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[i & MASK]);
                    const adp_v4f o4 = *reinterpret_cast<const LDS adp_v4f *>(&mybuf[(i - wv) & MASK]);
                    adp_v4f t4;
                    { const float d = a4.x - o4.x, p = o4.x - amean; amean += d * inv; t4.x = ((a4.x - amean) + p) * d; }
                    { const float d = a4.y - o4.y, p = o4.y - amean; amean += d * inv; t4.y = ((a4.y - amean) + p) * d; }
                    { const float d = a4.z - o4.z, p = o4.z - amean; amean += d * inv; t4.z = ((a4.z - amean) + p) * d; }
                    { const float d = a4.w - o4.w, p = o4.w - amean; amean += d * inv; t4.w = ((a4.w - amean) + p) * d; }
                    *reinterpret_cast<LDS adp_v4f *>(&myt[i - i0]) = t4;
                }
                for (; i < hi; i++) {
                    const float a = mybuf[i & MASK], o = mybuf[(i - wv) & MASK];
                    const float d = a - o, p = o - amean;
                    amean += d * inv;
                    myt[i - i0] = ((a - amean) + p) * d;
                }
            }
            SP_BARRIER();
        }
    } else if (wave == 3) {
        // ---- move_var, part 2, one chunk behind part 1: the clamped sum of squares and the output
        const int g = ln < SP_G ? ln : 0;
        const int n = ln < SP_G ? n_of[g] : 0;
        const float inv = (float)(1.0 / (double)wv);
        float s = 0.f;
        bool started = false;
        SP_BARRIER();                                   // P
        for (int k = 0; k < nchunks + 2; k++) {
            const int c = k - 1, i0 = c * SP_CH;
            if (c >= 0 && i0 < n) {
                const LDS float *myt = tbuf + (c & 1) * SP_G * SP_SO + g * SP_SO;
                LDS float *myout = outv + (c & 1) * SP_G * SP_SO + g * SP_SO;
                const int hi = n < i0 + SP_CH ? n : i0 + SP_CH;
                int i = i0 > wv ? i0 : wv;
                if (i < hi && !started) { s = s0[g]; started = true; }
                if (i == i0 && hi == i0 + SP_CH) {
                    // a whole chunk from registers, as in the mean wave.  (The clamp as max(ssq + t, 0) would shorten the chain from 18.5 to 13
                    // cycles per step -- tools/chain_latency.hip -- but drops a NaN the comparison keeps, and this wave is not the slowest.)
                    adp_v4f t4[SP_CH / 4];
#pragma unroll
                    for (int j = 0; j < SP_CH / 4; j++) t4[j] = *reinterpret_cast<const LDS adp_v4f *>(&myt[4 * j]);
#pragma unroll
                    for (int j = 0; j < SP_CH / 4; j++) {
                        sp_f2 s01, s23;
                        s += t4[j].x; if (s < 0) s = 0; s01.x = s;
                        s += t4[j].y; if (s < 0) s = 0; s01.y = s;
                        s += t4[j].z; if (s < 0) s = 0; s23.x = s;
                        s += t4[j].w; if (s < 0) s = 0; s23.y = s;
                        const sp_f2 r01 = s01 * (sp_f2){inv, inv}, r23 = s23 * (sp_f2){inv, inv};
                        *reinterpret_cast<LDS adp_v4f *>(&myout[4 * j]) = (adp_v4f){r01.x, r01.y, r23.x, r23.y};
                    }
                    i = hi;
                }
                for (; i + 4 <= hi; i += 4) {
                    const adp_v4f t4 = *reinterpret_cast<const LDS adp_v4f *>(&myt[i - i0]);
                    adp_v4f r4;
                    s += t4.x; if (s < 0) s = 0; r4.x = s * inv;
                    s += t4.y; if (s < 0) s = 0; r4.y = s * inv;
                    s += t4.z; if (s < 0) s = 0; r4.z = s * inv;
                    s += t4.w; if (s < 0) s = 0; r4.w = s * inv;
                    *reinterpret_cast<LDS adp_v4f *>(&myout[i - i0]) = r4;
                }
                for (; i < hi; i++) { s += myt[i - i0]; if (s < 0) s = 0; myout[i - i0] = s * inv; }
            }
            SP_BARRIER();
        }
        if (ln < SP_G && n > 0 && (s != s || (!started && s0[g] != s0[g]))) have[rid_of[g]] = 0;
    } else {
        // ---- storers: wave 4 the variances (two chunks behind part 1), wave 5 the means (one chunk behind): 16 lanes x 16 bytes per read
        const bool var = wave == 4;
        const int w = var ? wv : wm, lag = var ? 2 : 1;
        const LDS float *ob = var ? outv : outm;
        const int sub = ln & 15;
        GLB float *sp[NU];
        int nq[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int q = 4 * u + (ln >> 4);
            sp[u] = (GLB float *)series + (size_t)rid_of[q] * 2 * cap + (var ? cap : 0);
            nq[u] = n_of[q];
        }
        SP_BARRIER();                                   // P
        for (int k = 0; k < nchunks + 2; k++) {
            const int c = k - lag;
            if (c >= 0) {
                adp_v4f vv4[NU]; // (all of the chunk's LDS reads first: one wait, not one per store)
#pragma unroll
                for (int u = 0; u < NU; u++) vv4[u] = *reinterpret_cast<const LDS adp_v4f *>(ob + (c & 1) * SP_G * SP_SO + (4 * u + (ln >> 4)) * SP_SO + 4 * sub);
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    const int e = 4 * sub, i = c * SP_CH + e;
                    const adp_v4f v = vv4[u];
                    // series index of step i is i - (w - 1), defined for w - 1 <= i < n
#if defined(SP_ABL) && (SP_ABL & 2) // (timing only: no global stores)
                    if (v.x == 12345.678f)
#endif
                    if (i >= w - 1 && i + 3 < nq[u]) *reinterpret_cast<GLB sp_f4u *>(sp[u] + (i - w + 1)) = (sp_f4u){v.x, v.y, v.z, v.w};
                    else if (i + 3 >= w - 1 && i < nq[u]) { // (a vector wholly outside the series: the lanes of ended reads pass)
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; j++) if (i + j >= w - 1 && i + j < nq[u]) sp[u][i + j - w + 1] = vv[j];
                    }
                }
            }
            SP_BARRIER();
        }
    }
}
