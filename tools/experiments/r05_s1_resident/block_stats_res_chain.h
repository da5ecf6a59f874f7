// block_stats_res.h -- S1 calc_partition_stats of a LONG RNA partition with most of the segment kept ON CHIP between its two passes.
//
// k_partition_stats (block_stats.h) reads a segment twice from HBM -- pass A: numpy-ordered sum + bucket histogram, pass B: squared
// deviations + the copies for median and MAD -- with five 256-thread workgroups per CU: a gigabyte is in flight between a segment's
// passes and nothing on the chip holds a byte of it.  Here ONE persistent 512-thread workgroup per CU (256 registers per lane)
// walks the listed reads; of every read each wave KEEPS the first KR groups it streams in registers and the next KL in LDS, and
// pass B reads from HBM only what is left, last streamed first (one workgroup per CU keeps the footprint in flight under the 256 MB
// Infinity Cache).
//
// CHAIN LAYOUT.  A group = 1024 consecutive samples = eight of numpy's 128-sample leaves.  Lane (leaf ln >> 3, accumulator ln & 7) of
// a wave loads samples 128 leaf + 8 t + accumulator, t = 0 .. 15, straight from global memory (sixteen 4-byte loads whose lanes form
// 32-byte runs; tools/resident_bw.hip: within 3 % of the rate of 16-byte loads): exactly the sixteen terms of ONE of numpy's eight
// accumulator chains, in registers, in order.  No transposition through LDS, no wait inside a group: the chain is sixteen adds,
// the 8 accumulators and then the 8 leaves fold by lane shuffles in numpy's pairing -- the sums are those of block_np_sum bit for
// bit -- and the side effects (histogram cell; bracket tests) apply to the registers.  The ragged chunk behind the whole ones
// (< 8192 samples, <= 64 leaves of numpy's pairwise recursion) goes the same way: one group of eight ragged leaves per wave.
// Pass B's copies (the median's bucket, the MAD bracket: ~2 % of the samples) take no atomics: every wave fills regions of its own
// behind a cursor it keeps in a register (ballot + prefix count); the regions are packed together before the selections.
//
// The serial phases (bucket search and MAD prediction between the passes; the selections behind pass B) would leave the CU's
// memory pipes idle with a single workgroup on it: pass B's first groups are requested BEFORE the phase between the passes, and
// the NEXT read's resident groups are requested before the selections -- into the registers pass B has just emptied -- so that the
// CU streams through them.  Nothing is CALLED while groups sit in registers (a call would spill them): the helpers are the
// inlined forms.  Only the fast path lives here.  Whatever needs another look at the segment (a NaN, a window that missed, a
// bucket of many values, the lower median on a bucket's first sample, a MAD bracket that does not prove itself, a region that
// overflows) is put on a list and left to k_partition_stats in its list mode: rare, slower there, never different.
//
// reference: adapted/partition/signal_partitions.py:81-96 (np.mean, np.std, np.median, np.median(|x - med|) of signal[polya_end:]).
#pragma once
#include "block_stats.h"

#define RS_THREADS 512
#define RS_NW (RS_THREADS / 64)
#define RS_LONG_MIN 65536 // shortest RNA partition taken (samples): one numpy chunk per wave
#define RS_BCAP 384       // a wave's region of MAD-bracket samples (RS_NW regions = the histogram's 3072 cells in size)
#define RS_MCAP 64        // a wave's region of samples of the median's bucket

template <class ROW> struct ResShape;
template <> struct ResShape<RowF32> { enum { KR = 8, KL = 3, PF = 2 }; };

typedef BlockScratchT<RS_THREADS, true, BS_BINS + 64> ResScratch;

template <int KL>
struct ResShared {
    ResScratch bs;
    float slabsum[BS_MAXCHUNK * 8]; // sums of the whole groups, by group index (numpy chunk c = groups 8c .. 8c + 7)
    float madreg[RS_NW * RS_BCAP];  // pass B: the waves' regions of bracket samples ...
    float medreg[RS_NW * RS_MCAP];  // ... and of samples of the median's bucket
    int item_next, overflow, pad_[2];
    adp_v4f lres[RS_NW * KL * 4 * 64]; // the LDS-resident groups, lane-major: [wave][slot][u][lane] = c[4u .. 4u + 3]
};

// the sixteen chain terms of this lane for the group that starts at sample `pos`
template <bool CACHED>
static __device__ __forceinline__ void rs_load(const RowF32 &x, long long pos, int ln, float (&c)[16])
{
    const GLB float *q = x.p + pos + (ln >> 3) * 128 + (ln & 7);
#pragma unroll
    for (int t = 0; t < 16; t++) c[t] = CACHED ? q[8 * t] : __builtin_nontemporal_load(q + 8 * t);
}

// what a pass carries from group to group (registers)
struct RsPassA { uint32_t wlo; };
struct RsPassB {
    float mean;
    SideParam p;
    uint32_t cnt_lt;      // samples closer to the bucket centre than the bracket (this lane's)
    int bcur, mcur;       // cursors of the wave's regions (wave-uniform)
    uint32_t kmin, kmax;  // keys copied out of the median's bucket (this lane's)
};

static __device__ __forceinline__ int rs_prefix(unsigned long long m) // lanes of m below this one
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// pass B's side effect for one sample (uniform control flow: ballots and scalar branches)
static __device__ __forceinline__ void rs_collect(float v, bool valid, RsPassB &b, LDS float *breg, LDS float *mreg)
{
    const float dt = fabsf(v - b.p.c);
    const bool lt = valid && dt < b.p.P;
    b.cnt_lt += lt ? 1u : 0u;
    const bool inbr = valid && !lt && dt <= b.p.Q;
    const unsigned long long mb = __ballot(inbr);
    if (mb) {
        const int pos = b.bcur + rs_prefix(mb);
        if (inbr && pos < RS_BCAP) breg[pos] = v;
        b.bcur += __popcll(mb);
    }
    const bool cand = valid && dt <= b.p.hw; // (a superset of the median's bucket)
    if (__ballot(cand)) {
        const uint32_t key = f2key(v);
        const bool isb = cand && (key >> BS_KSH) == b.p.key;
        const unsigned long long mm = __ballot(isb);
        if (mm) {
            const int pos = b.mcur + rs_prefix(mm);
            if (isb) {
                if (pos < RS_MCAP) mreg[pos] = v;
                b.kmin = key < b.kmin ? key : b.kmin;
                b.kmax = key > b.kmax ? key : b.kmax;
            }
            b.mcur += __popcll(mm);
        }
    }
}

// the 8 accumulators of a leaf (lanes 8 l .. 8 l + 7), in numpy's pairing ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
static __device__ __forceinline__ float rs_fold8(float r)
{
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    return r;
}

// A WHOLE group: every lane returns its sum (three more levels of numpy's balanced tree over the eight leaves).
template <int PASS, class BS>
static __device__ __forceinline__ float rs_group(const float (&c)[16], RsPassA &a, RsPassB &b, LDS BS *bs, LDS float *breg, LDS float *mreg)
{
    float r;
    if (PASS == 0) {
        r = c[0];
#pragma unroll
        for (int t = 1; t < 16; t++) r += c[t];
        LDS uint32_t *bins = bs_bins(bs);
#pragma unroll
        for (int t = 0; t < 16; t++) __hip_atomic_fetch_add(&bins[bs_cell(c[t], a.wlo)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        { const float d = c[0] - b.mean; r = d * d; }
#pragma unroll
        for (int t = 1; t < 16; t++) { const float d = c[t] - b.mean; r += d * d; }
#pragma unroll
        for (int t = 0; t < 16; t++) rs_collect(c[t], true, b, breg, mreg);
    }
    r = rs_fold8(r);
    r = r + __shfl_xor(r, 8);
    r = r + __shfl_xor(r, 16);
    r = r + __shfl_xor(r, 32);
    return r;
}

// A group of eight leaves of the RAGGED chunk (numpy's pairwise recursion: bs_tail_leaves): lane (l, j) has the chain terms
// c[t] = leaf[8 t + j], t < nt = len / 8, and e = leaf[8 nt + j] for j < k = len % 8 (len < 8: nt = 0, k = len).  numpy sums a leaf of
// at least 8 samples by its 8 accumulators and adds the k remaining samples in sequence; a shorter one from zero in sequence.
// Lane j == 0 of each leaf stores the leaf's sum by tree slot.
struct RsRagged { float c[16]; float e; int nt, k, slot; };
template <bool CACHED, class BS>
static __device__ __forceinline__ void rs_ragged_load(const RowF32 &xt, int tail, int grp, int ln, LDS BS *bs, RsRagged &g)
{
    const int L = 8 * grp + (ln >> 3), j = ln & 7;
    const bool have = L < bs->nleaf;
    const int off = have ? bs->leaf_off[L] : 0, len = have ? bs->leaf_len[L] : 0;
    g.nt = len >> 3; g.k = len & 7;
    g.slot = have ? bs->leaf_slot[L] : -1;
    const int last = tail - 1;
    const GLB float *q = xt.p;
#pragma unroll
    for (int t = 0; t < 16; t++) { const int i = off + 8 * t + j; const int ii = i < last ? i : last; g.c[t] = CACHED ? q[ii] : __builtin_nontemporal_load(q + ii); }
    { const int i = off + 8 * g.nt + j; const int ii = i < last ? i : last; g.e = CACHED ? q[ii] : __builtin_nontemporal_load(q + ii); }
}
template <int PASS, class BS>
static __device__ __forceinline__ void rs_ragged_group(const RsRagged &g, RsPassA &a, RsPassB &b, LDS BS *bs, LDS float *breg, LDS float *mreg)
{
    const int ln = threadIdx.x & 63, j = ln & 7;
    const bool ev = j < g.k;
    float r = 0.0f;
    if (PASS == 0) {
        r = g.c[0];
#pragma unroll
        for (int t = 1; t < 16; t++) r = t < g.nt ? r + g.c[t] : r;
        LDS uint32_t *bins = bs_bins(bs);
#pragma unroll
        for (int t = 0; t < 16; t++)
            if (t < g.nt) __hip_atomic_fetch_add(&bins[bs_cell(g.c[t], a.wlo)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (ev) __hip_atomic_fetch_add(&bins[bs_cell(g.e, a.wlo)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        { const float d = g.c[0] - b.mean; r = d * d; }
#pragma unroll
        for (int t = 1; t < 16; t++) { const float d = g.c[t] - b.mean; r = t < g.nt ? r + d * d : r; }
#pragma unroll
        for (int t = 0; t < 16; t++) rs_collect(g.c[t], t < g.nt, b, breg, mreg);
        rs_collect(g.e, ev, b, breg, mreg);
    }
    r = rs_fold8(r);
    float ex = g.e;
    if (PASS == 1) { const float d = g.e - b.mean; ex = d * d; }
    float res = g.nt > 0 ? r : 0.0f;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const float v = __shfl(ex, (ln & ~7) + i);
        res = i < g.k ? res + v : res;
    }
    if (j == 0 && g.slot >= 0) bs->tleaf[g.slot] = res;
}

#ifdef ADP_PHASE_TIMING
// (debug build) 10 ns ticks of the constant clock per phase, summed over the reads by each workgroup's first thread
__device__ unsigned long long g_res_phase[16];
#define RS_PHASE(slot) do { const long long t_ = wall_clock64(); if (tid == 0) atomicAdd(&g_res_phase[slot], (unsigned long long)(t_ - tph_)); tph_ = t_; } while (0)
#else
#define RS_PHASE(slot) do { } while (0)
#endif

template <class SIG>
__global__ void __launch_bounds__(RS_THREADS) k_partition_rna_res(SIG sigs, int m, const PartReq *__restrict__ req, adp_row *__restrict__ rows,
                                                                  const int *__restrict__ list, ResCounters *__restrict__ cnt, int *__restrict__ redo)
{
    typedef typename SIG::Row X;
    constexpr int KR = ResShape<X>::KR, KL = ResShape<X>::KL, PF = ResShape<X>::PF, NW = RS_NW;
    typedef ResShared<KL> Sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_mem_[];
    LDS Sh *sh = (LDS Sh *)rs_mem_;
    LDS ResScratch *bs = &sh->bs;
    int tid = threadIdx.x, ln = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave-uniform: its group counts and positions stay scalar)
    LDS float *breg = sh->madreg + w * RS_BCAP, *mreg = sh->medreg + w * RS_MCAP;
    const int count = cnt->n_list;
    int cur = blockIdx.x;
    if (cur >= count) return;
    int nxt = cur + (int)gridDim.x; // the first two items are dealt statically, the later ones by the counter
    if (tid == 0) { bs->tail_cached = -1; sh->overflow = 0; }
    __syncthreads();

    float keep[KR][16];
    // the RNA partition of a listed read: signal[min(polya_end, S) : S]
    auto describe = [&](int item, X &xo, int &no, int &ro) {
        ro = list[item];
        const long long S = req[ro].S, pe = req[ro].p_e;
        const long long a = pe < S ? pe : S;
        no = (int)(S - a);
        xo = sigs.row(ro, m) + a;
    };
    // the three pivot samples, then this wave's first KR groups (group w + NW q of the segment).  Every load is issued whatever the
    // length (a group that does not exist reads group w again): straight-line code, so that the waits on these registers are counted
    auto request = [&](const X &xo, int no, float (&pv)[3]) {
        pv[0] = xo.p[no / 4]; pv[1] = xo.p[no / 2]; pv[2] = xo.p[(3 * (long long)no) / 4];
        const int ngrp = (no / 8192) * 8;
#pragma unroll
        for (int q = 0; q < KR; q++) {
            const int g = w + NW * q;
            rs_load<false>(xo, (long long)(g < ngrp ? g : w) * 1024, ln, keep[q]);
        }
    };

    X x, xn;
    int n, r, nn, rn;
    float pv[3], pvn[3];
    describe(cur, x, n, r);
    request(x, n, pv);
#ifdef ADP_PHASE_TIMING
    long long tph_ = wall_clock64();
#endif
    for (;;) {
        tid = bs_tid<ResScratch>(); ln = tid & 63; // (per read: nothing derived from the index is kept across the loop)
        const int nchunk = n / 8192, ngrp = nchunk * 8;
        const int mine = ngrp > w ? (ngrp - w + NW - 1) / NW : 0; // whole groups of this wave: w, w + NW, ...
        const int nstream = mine > KR ? mine - KR : 0;            // ... of which those behind the first KR are streamed
        const int tail = n - nchunk * 8192;
        const X xtail = x + (long long)nchunk * 8192;
        const int k1 = n / 2;
        auto gpos = [&](int q) { return (long long)(w + NW * q) * 1024; };
        // ---- pass A: numpy-ordered sum + bucket histogram -------------------------------------------------
        RsPassA pa;
        RsPassB pb;
        {
            const float pivot = fmaxf(fminf(pv[0], pv[1]), fminf(fmaxf(pv[0], pv[1]), pv[2]));
            const uint32_t kb = f2key(pivot) >> BS_KSH;
            pa.wlo = kb >= BS_BINS / 2 ? kb - BS_BINS / 2 : 0u; // centred on the pivot
        }
        for (int i = tid; i < BS_BINS + 4; i += RS_THREADS) bs->hist[i] = 0;
        if (tail > 0) bs_tail_leaves(tail, bs); // (barriers inside)
        __syncthreads();
        RS_PHASE(0);
        float pf[PF][16];
#pragma unroll
        for (int d = 0; d < PF; d++)
            if (d < nstream) rs_load<true>(x, gpos(KR + d), ln, pf[d]);
        RsRagged rg;
        if (tail > 0) rs_ragged_load<true>(xtail, tail, w, ln, bs, rg);
#pragma unroll
        for (int q = 0; q < KR; q++)
            if (q < mine) {
                const float s_ = rs_group<0>(keep[q], pa, pb, bs, breg, mreg);
                if (ln == 0) sh->slabsum[w + NW * q] = s_;
            }
        RS_PHASE(1);
        for (int i = 0; i < nstream; i++) {
            float c[16];
#pragma unroll
            for (int t = 0; t < 16; t++) c[t] = pf[0][t];
#pragma unroll
            for (int d = 0; d + 1 < PF; d++)
#pragma unroll
                for (int t = 0; t < 16; t++) pf[d][t] = pf[d + 1][t];
            if (i + PF < nstream) rs_load<true>(x, gpos(KR + i + PF), ln, pf[PF - 1]);
            if (i < KL) {
#pragma unroll
                for (int u = 0; u < 4; u++) { const adp_v4f t4 = {c[4 * u], c[4 * u + 1], c[4 * u + 2], c[4 * u + 3]}; sh->lres[((w * KL + i) * 4 + u) * 64 + ln] = t4; }
            }
            const float s_ = rs_group<0>(c, pa, pb, bs, breg, mreg);
            if (ln == 0) sh->slabsum[w + NW * (KR + i)] = s_;
        }
        RS_PHASE(2);
        // (a wave's first ragged group was requested at the top of the pass; a ragged chunk of more than 64 leaves has more)
        const int nrg = tail > 0 ? (bs->nleaf + 7) >> 3 : 0;
        if (tail > 0) {
            rs_ragged_group<0>(rg, pa, pb, bs, breg, mreg);
            for (int g = w + NW; g < nrg; g += NW) { rs_ragged_load<true>(xtail, tail, g, ln, bs, rg); rs_ragged_group<0>(rg, pa, pb, bs, breg, mreg); }
        }
        RS_PHASE(3);
        // pass B's first streamed groups (the last ones of pass A) and its ragged group are requested now: they arrive during the
        // phase between the passes
        const int nsb = nstream > KL ? nstream - KL : 0; // groups pass B streams: KR + KL .. mine - 1, walked from the end
#pragma unroll
        for (int d = 0; d < PF; d++)
            if (d < nsb) rs_load<false>(x, gpos(KR + nstream - 1 - d), ln, pf[d]);
        if (tail > 0) rs_ragged_load<false>(xtail, tail, w, ln, bs, rg);
        __syncthreads();
        // numpy's order above the groups: a chunk's eight group sums pairwise, the chunks in sequence, the ragged chunk last
        auto fold = [&]() {
            if (tid < nchunk) {
                const LDS float *t = sh->slabsum + 8 * tid;
                bs->chunk_sum[tid] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            }
            float ragged = 0.0f;
            if (tail > 0 && tid < 64) ragged = bs_tail_tree(bs); // (lane 0 holds the root)
            __syncthreads();
            if (tid == 0) {
                float total = 0.0f;
                for (int ch = 0; ch < nchunk; ch++) total += bs->chunk_sum[ch];
                if (tail > 0) total += ragged;
                bs->bcast[0] = total;
            }
            __syncthreads();
            return bs->bcast[0];
        };
        const float mean = fold() / (float)n;
        const uint32_t below = bs->hist[0];
        bool bail = mean != mean; // a NaN, or infinities of both signs: one look at the segment decides -- in the list mode
        int bin = 0, rk = 0;
        float c = 0.f, w0 = 0.f, P = 0.f, Q = 0.f, hw = 0.f;
        if (!bail) {
            block_find_bin<BS_BINS>(bs, bs_bins(bs), k1, (int)below);
            bail = bs->flag != 0;
            bin = bs->bin; rk = k1 - bs->before;
            __syncthreads();
        }
        if (!bail) {
            const uint32_t key_lo = (pa.wlo + (uint32_t)bin) << BS_KSH;
            const float c_lo = key2f(key_lo), c_hi = key2f(key_lo + (1u << BS_KSH));
            c = 0.5f * (c_lo + c_hi);
            w0 = c_hi - c_lo;
            if (!(w0 > 0.f) || __builtin_isinf(c_hi) || __builtin_isinf(c_lo)) bail = true;
            else {
                const float hm = fmaxf(c - c_lo, c_hi - c) * 1.0001f;
                hw = __uint_as_float(__float_as_uint(hm) + 2u);
            }
        }
        if (!bail) bail = !bs_predict_mad_i(bs, pa.wlo, k1, c, w0, P, Q);
        __syncthreads();
        // ---- pass B: squared deviations + the median's bucket + the MAD bracket -----------------------------
        pb.mean = mean;
        pb.p.key = pa.wlo + (uint32_t)bin; pb.p.c = c; pb.p.hw = hw; pb.p.do_mad = 1; pb.p.P = P; pb.p.Q = Q;
        pb.cnt_lt = 0; pb.bcur = 0; pb.mcur = 0; pb.kmin = 0xffffffffu; pb.kmax = 0u;
        RS_PHASE(4);
        if (!bail) {
            if (tail > 0) {
                rs_ragged_group<1>(rg, pa, pb, bs, breg, mreg);
                for (int g = w + NW; g < nrg; g += NW) { rs_ragged_load<false>(xtail, tail, g, ln, bs, rg); rs_ragged_group<1>(rg, pa, pb, bs, breg, mreg); }
            }
            RS_PHASE(5);
            for (int jb = 0; jb < nsb; jb++) { // streamed groups, last first
                const int i = nstream - 1 - jb;
                float cc[16];
#pragma unroll
                for (int t = 0; t < 16; t++) cc[t] = pf[0][t];
#pragma unroll
                for (int d = 0; d + 1 < PF; d++)
#pragma unroll
                    for (int t = 0; t < 16; t++) pf[d][t] = pf[d + 1][t];
                if (jb + PF < nsb) rs_load<false>(x, gpos(KR + i - PF), ln, pf[PF - 1]);
                const float s_ = rs_group<1>(cc, pa, pb, bs, breg, mreg);
                if (ln == 0) sh->slabsum[w + NW * (KR + i)] = s_;
            }
            RS_PHASE(6);
            const int nl = nstream < KL ? nstream : KL;
            for (int i = 0; i < nl; i++) { // the groups kept in LDS
                float cc[16];
#pragma unroll
                for (int u = 0; u < 4; u++) { const adp_v4f t4 = sh->lres[((w * KL + i) * 4 + u) * 64 + ln]; cc[4 * u] = t4.x; cc[4 * u + 1] = t4.y; cc[4 * u + 2] = t4.z; cc[4 * u + 3] = t4.w; }
                const float s_ = rs_group<1>(cc, pa, pb, bs, breg, mreg);
                if (ln == 0) sh->slabsum[w + NW * (KR + i)] = s_;
            }
            RS_PHASE(7);
#pragma unroll
            for (int q = 0; q < KR; q++) // the groups kept in registers
                if (q < mine) {
                    const float s_ = rs_group<1>(keep[q], pa, pb, bs, breg, mreg);
                    if (ln == 0) sh->slabsum[w + NW * q] = s_;
                }
        }
        // ---- the registers are free: the next read's resident groups are requested before the selections ---------------
        RS_PHASE(8);
        unsigned int ticket = 0;
        if (tid == 0) ticket = atomicAdd(&cnt->work, 1u);
        const bool have_next = nxt < count;
        describe(have_next ? nxt : cur, xn, nn, rn);
        request(xn, nn, pvn);
        RS_PHASE(9);
        float sd = 0.f, med = 0.f, mad = 0.f;
        int M = 0, ncol = 0;
        if (!bail) {
            // the waves' counts, cursors and key ranges meet in LDS; the regions are packed together: bracket samples to the histogram's
            // storage, the bucket's behind them (where block_segment_stats keeps them)
            const uint32_t w2 = (uint32_t)wave_sum((int)pb.cnt_lt);
            const uint32_t kmn = wave_min(pb.kmin), kmx = wave_max(pb.kmax);
            if (tid == 0) { bs->cntb = 0; bs->kmin = 0xffffffffu; bs->kmax = 0u; }
            __syncthreads();
            if (ln == 0) {
                if (w2) __hip_atomic_fetch_add(&bs->cntb, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_min(&bs->kmin, kmn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&bs->kmax, kmx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                bs->scan[w] = pb.bcur; bs->scan[NW + w] = pb.mcur;
                if (pb.bcur > RS_BCAP) sh->overflow = 1;
            }
            __syncthreads();
            int boff[NW + 1], moff[NW + 1];
            boff[0] = 0; moff[0] = 0;
#pragma unroll
            for (int i = 0; i < NW; i++) { boff[i + 1] = boff[i] + bs->scan[i]; moff[i + 1] = moff[i] + bs->scan[NW + i]; }
            M = boff[NW]; ncol = moff[NW];
            const bool over = sh->overflow != 0;
            __syncthreads();
            if (over) { bail = true; if (tid == 0) sh->overflow = 0; }
            else {
                LDS float *dstb = (LDS float *)bs->hist;
                if (M <= BS_MADCAP) {
#pragma unroll
                    for (int i = 0; i < NW; i++) {
                        const int cn = boff[i + 1] - boff[i];
                        if (tid < cn) dstb[boff[i] + tid] = sh->madreg[i * RS_BCAP + tid]; // (RS_BCAP <= RS_THREADS)
                    }
                }
                LDS float *dstm = bs_collect(bs);
#pragma unroll
                for (int i = 0; i < NW; i++) {
                    const int cn = moff[i + 1] - moff[i];
                    const int cc = cn < RS_MCAP ? cn : RS_MCAP;
                    if (tid < cc && moff[i] + tid < BS_MEDCAP) dstm[moff[i] + tid] = sh->medreg[i * RS_MCAP + tid];
                }
                // (a bucket list that overflowed a wave's region is only used when the bucket holds ONE value: below)
            }
            __syncthreads();
            sd = sqrtf(fold() / (float)n);
        }
        // ---- median inside its bucket ----------------------------------------------------------------------
        if (!bail) {
            const uint32_t kmin = bs->kmin, kmax = bs->kmax;
            const bool first_of_even = (n & 1) == 0 && rk == 0; // the lower median is the largest sample BELOW the bucket: another pass
            bool region_over = false;
#pragma unroll
            for (int i = 0; i < NW; i++) region_over |= bs->scan[NW + i] > RS_MCAP;
            __syncthreads();
            if (first_of_even) bail = true;
            else if (ncol > BS_MEDCAP || region_over) {
                if (kmin == kmax) { // (quantised data: the bucket holds one value)
                    const float vk = key2f(kmin);
                    med = vk;
                    if ((n & 1) == 0) med = (vk + vk) / 2.0f;
                } else bail = true;
            } else med = bs_median_from_bucket_i(bs, bs_collect(bs), ncol, n, rk, 0u);
        }
        // ---- MAD inside the bracket, if that can be proven ---------------------------------------------------
        if (!bail) {
            const int rel = k1 - (int)bs->cntb;
            const bool need_prev = (n & 1) == 0;
            __syncthreads();
            if (M <= BS_MADCAP && rel >= (need_prev ? 1 : 0) && rel < M) {
                float vk, vkm1;
                block_select2_lds_i((const LDS float *)bs->hist, M, rel, 1, med, bs, vk, vkm1);
                if (tid == 0) {
                    const float lo = need_prev ? vkm1 : vk;
                    const float dm = fabsf(med - c) + 0.25f * w0; // (see block_segment_stats)
                    const bool proven = (lo >= P + dm) && (vk <= Q - dm) && (dm < 2.0f * w0);
                    bs->bcast[2] = need_prev ? (vkm1 + vk) / 2.0f : vk;
                    bs->flag = proven ? 1 : 0;
                }
                __syncthreads();
                if (bs->flag) mad = bs->bcast[2]; else bail = true;
            } else bail = true;
        }
        if (tid == 0) {
            if (!bail) {
                adp_row *row = rows + r;
                row->col[ADP_C_RNA_LEN] = (double)n;
                row->col[ADP_C_RNA_MEAN] = (double)mean;
                row->col[ADP_C_RNA_STD] = (double)sd;
                row->col[ADP_C_RNA_MED] = (double)med;
                row->col[ADP_C_RNA_MAD] = (double)mad;
                row->present |= 31ull << ADP_C_RNA_LEN;
                if (!(g_ablate & 262144)) {
                    unsigned long long *tl = g_bs_tally[blockIdx.x & (ADP_NTALLY - 1)];
                    atomicAdd(&tl[0], 1ull);
                    atomicAdd(&tl[1], 1ull);
                    atomicAdd(&tl[5], 1ull); // finished by this kernel
                }
            } else {
                redo[atomicAdd(&cnt->n_redo, 1)] = r;
                if (!(g_ablate & 262144)) atomicAdd(&g_bs_tally[blockIdx.x & (ADP_NTALLY - 1)][6], 1ull);
            }
            sh->item_next = (int)ticket + 2 * (int)gridDim.x;
        }
        __syncthreads();
        RS_PHASE(10);
        if (!have_next) break;
        cur = nxt; x = xn; n = nn; r = rn;
        pv[0] = pvn[0]; pv[1] = pvn[1]; pv[2] = pvn[2];
        nxt = sh->item_next;
        __syncthreads();
    }
}
