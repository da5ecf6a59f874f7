// s1_resident.h -- the two streaming passes of S1 calc_partition_stats over a LONG RNA partition with part of the segment kept ON
// CHIP between them; everything behind the passes stays with k_partition_stats (block_stats.h), which takes over from a record.
//
// k_partition_stats reads a segment twice from HBM -- pass A: numpy-ordered sum + bucket histogram, pass B: squared deviations +
// the copies for median and MAD -- with five 256-thread workgroups per CU: a gigabyte is in flight between a segment's passes and
// nothing on the chip holds a byte of it.  Here ONE persistent 512-thread workgroup per CU (256 registers per lane) walks the
// listed reads; of every read each wave KEEPS the first KR groups it streams in registers and the next KL in LDS, and pass B
// reads from HBM only what is left, last streamed first (one workgroup per CU keeps the footprint in flight under the 256 MB
// Infinity Cache).  Two earlier forms of this kernel lost to the two-pass kernel (profiles/r05_tried_and_dropped.txt: with eight
// waves per CU every instruction and every LDS round trip of a group's body is exposed); this one differs in what that record names:
//
// CHAIN LAYOUT.  A group = 1024 consecutive samples = eight of numpy's 128-sample leaves.  Lane (leaf ln >> 3, accumulator ln & 7)
// loads samples 128 leaf + 8 t + accumulator, t = 0 .. 15, straight from global memory (sixteen 4-byte loads whose lanes form
// 32-byte runs: within 3 % of the rate of 16-byte loads, tools/resident_bw.hip): the sixteen terms of ONE of numpy's eight
// accumulator chains, in registers, in order.  No transposition through LDS and no wait inside a group: sixteen adds, then the
// 8 accumulators and the 8 leaves fold by DPP row shifts in numpy's pairing (no ds_bpermute behind the histogram's atomics) --
// the sums are block_np_sum's bit for bit.  The ragged chunk behind the whole ones (< 8192 samples) goes the same way, a group of
// eight ragged leaves per wave.
// PASS B WITHOUT ATOMICS OR BRANCHES PER SAMPLE.  Per sample: the squared deviation, |x - c| against the bracket (a carry-add
// counts the closer ones), and ONE wave mask of the special samples -- in the MAD bracket or within the bucket's half width of its
// centre, ~1 % -- which a prefix count turns into slots of a region the wave owns (cursor in a scalar register).  The regions are
// packed into the read's record; k_partition_stats sorts them into its two lists exactly as its own pass B does.
// NOTHING SERIAL BEHIND PASS B.  The selections (median in its bucket, MAD in its bracket, their fallbacks) are 11-18 us of barriers
// for one workgroup alone on a CU; they run in k_partition_stats, five workgroups per CU, from the record (S1Rec, block_stats.h).
// What stays between the passes (fold, bucket search, MAD prediction) has pass B's first groups in flight; the next read's
// resident groups are requested as soon as pass B has emptied the registers.
// Nothing is CALLED while groups sit in registers (a call would spill them): block_find_bin / bs_predict_mad in their inlined forms.
//
// reference: adapted/partition/signal_partitions.py:81-96 (np.mean, np.std of signal[polya_end:]; np.median and the MAD from the
// record's special samples in k_partition_stats).
#pragma once
#include "block_stats.h"

#define RS_THREADS 512
#define RS_NW (RS_THREADS / 64)
#define RS_LONG_MIN 65536 // shortest RNA partition taken (samples): one numpy chunk per wave
static_assert(RS_NW == 8, "S1_REC_FLOATS counts eight waves' regions");

template <class ROW> struct ResShape;
template <> struct ResShape<RowF32> { enum { KR = 8, KL = 3 }; };

typedef BlockScratchT<RS_THREADS, true, BS_BINS + 64> ResScratch;

template <int KL>
struct ResShared {
    ResScratch bs;
    float slabsum[BS_MAXCHUNK * 8];          // sums of the whole groups, by group index (numpy chunk c = groups 8c .. 8c + 7)
    float special[RS_NW * S1_SPECIAL_CAP];   // pass B: the waves' regions of special samples
    float dump[64];                          // where the lanes without a special sample write
    int cursor[RS_NW];
    int item_next, pad_[3];
    adp_v4f lres[RS_NW * KL * 4 * 64];       // the LDS-resident groups, lane-major: [wave][slot][u][lane] = c[4u .. 4u + 3]
};

// ---- list of the reads with a long RNA partition: list[k] = read, rec_of[read] = k (or -1); one atomic per wave
__global__ void __launch_bounds__(256) k_s1_list(const PartReq *__restrict__ req, int n_reads, int long_min, int *__restrict__ list,
                                                 int *__restrict__ rec_of, int *__restrict__ n_list)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    bool take = false;
    if (r < n_reads) {
        const PartReq q = req[r];
        const long long S = q.S, a = q.p_e < S ? q.p_e : S;
        take = q.valid && !q.p_none && q.p_e >= 0 && S - a >= long_min;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(n_list, __builtin_popcountll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    const int at = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (r < n_reads) rec_of[r] = take ? at : -1;
    if (take) list[at] = r;
}

// the sixteen chain terms of this lane for the group that starts at sample `pos`
template <bool CACHED>
static __device__ __forceinline__ void rs_load(const RowF32 &x, long long pos, int ln, float (&c)[16])
{
    const GLB float *q = x.p + pos + (ln >> 3) * 128 + (ln & 7);
#pragma unroll
    for (int t = 0; t < 16; t++) c[t] = CACHED ? q[8 * t] : __builtin_nontemporal_load(q + 8 * t);
}

// v + (the value n lanes up in the lane's row of 16): DPP row_shl, lanes shifted in from beyond the row read 0
template <int N>
static __device__ __forceinline__ float rs_add_shl(float v)
{
    const int up = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, up);
}
// the 8 accumulators of a leaf (lanes 8 l .. 8 l + 7) in numpy's pairing ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)): valid in lane 8 l
static __device__ __forceinline__ float rs_fold8(float r)
{
    r = rs_add_shl<1>(r);
    r = rs_add_shl<2>(r);
    r = rs_add_shl<4>(r);
    return r;
}
// ... and the group's eight leaves, three more levels of numpy's balanced tree: every lane returns the group's sum
static __device__ __forceinline__ float rs_fold_group(float r)
{
    r = rs_add_shl<8>(rs_fold8(r)); // lane 0 of row k: leaves 2k and 2k + 1
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 48));
    return (a + b) + (c + d);
}

// what pass B carries from group to group
struct RsPassB {
    float mean, c, P, Q, hw;
    uint32_t cnt_lt;            // samples with |x - c| < P (this lane's)
    int cur;                    // entries of the wave's region (wave-uniform; beyond S1_SPECIAL_CAP: overflow)
    LDS float *region, *dump;   // the wave's region; this lane's dump slot
};

// pass B's side effect for one sample: no branch, no atomic.  The three comparisons leave wave masks in scalar registers
// (__builtin_amdgcn_fcmpf), the special samples' mask is two scalar operations on them, a lane's slot is the cursor plus the mask's
// bits below the lane.  valid: a wave mask of the lanes whose sample exists (all of them in a whole group).
static __device__ __forceinline__ void rs_collect(float v, unsigned long long valid, RsPassB &b)
{
    const float ae = fabsf(v - b.c);
    const unsigned long long lt = __builtin_amdgcn_fcmpf(ae, b.P, 4 /* ordered < */) & valid;
    const unsigned long long leq = __builtin_amdgcn_fcmpf(ae, b.Q, 5 /* ordered <= */) & valid;
    const unsigned long long lehw = __builtin_amdgcn_fcmpf(ae, b.hw, 5) & valid;
    const unsigned long long sp = (leq & ~lt) | lehw; // in the MAD bracket (not closer than P, not farther than Q) or within the bucket's half width
    b.cnt_lt += __builtin_amdgcn_inverse_ballot_w64(lt) ? 1u : 0u;
    int pos = b.cur + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(sp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sp, 0u));
    pos = pos < S1_SPECIAL_CAP - 1 ? pos : S1_SPECIAL_CAP - 1; // (an overflowing region piles up in its last slot; the cursor tells)
    LDS float *dst = __builtin_amdgcn_inverse_ballot_w64(sp) ? b.region + pos : b.dump;
    *dst = v;
    b.cur += __builtin_popcountll(sp);
}

template <int PASS, class BS>
static __device__ __forceinline__ float rs_group(const float (&c)[16], uint32_t wlo, RsPassB &b, LDS BS *bs)
{
    float r;
    if (PASS == 0) {
        r = c[0];
#pragma unroll
        for (int t = 1; t < 16; t++) r += c[t];
        LDS uint32_t *bins = bs_bins(bs);
#pragma unroll
        for (int t = 0; t < 16; t++) __hip_atomic_fetch_add(&bins[bs_cell(c[t], wlo)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        { const float d = c[0] - b.mean; r = d * d; }
#pragma unroll
        for (int t = 1; t < 16; t++) { const float d = c[t] - b.mean; r += d * d; }
#pragma unroll
        for (int t = 0; t < 16; t++) rs_collect(c[t], ~0ull, b);
    }
    return rs_fold_group(r);
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
static __device__ __forceinline__ void rs_ragged_group(const RsRagged &g, uint32_t wlo, RsPassB &b, LDS BS *bs)
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
            if (t < g.nt) __hip_atomic_fetch_add(&bins[bs_cell(g.c[t], wlo)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (ev) __hip_atomic_fetch_add(&bins[bs_cell(g.e, wlo)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        { const float d = g.c[0] - b.mean; r = d * d; }
#pragma unroll
        for (int t = 1; t < 16; t++) { const float d = g.c[t] - b.mean; r = t < g.nt ? r + d * d : r; }
#pragma unroll
        for (int t = 0; t < 16; t++) rs_collect(g.c[t], __builtin_amdgcn_ballot_w64(t < g.nt), b);
        rs_collect(g.e, __builtin_amdgcn_ballot_w64(ev), b);
    }
    r = rs_fold8(r); // (valid in lane j == 0)
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

// grid = one workgroup per CU (at most the listed reads); dynamic LDS = sizeof(ResShared<KL>).  work: the item counter (zeroed by the host)
template <class SIG>
__global__ void __launch_bounds__(RS_THREADS) k_partition_rna_res(SIG sigs, int m, const PartReq *__restrict__ req, const int *__restrict__ list,
                                                                  const int *__restrict__ n_list, unsigned int *__restrict__ work, float *__restrict__ recs)
{
    typedef typename SIG::Row X;
    constexpr int KR = ResShape<X>::KR, KL = ResShape<X>::KL, NW = RS_NW;
    typedef ResShared<KL> Sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_mem_[];
    LDS Sh *sh = (LDS Sh *)rs_mem_;
    LDS ResScratch *bs = &sh->bs;
    int tid = threadIdx.x, ln = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave-uniform: its group counts and positions stay scalar)
    const int count = *n_list;
    int cur = blockIdx.x;
    if (cur >= count) return;
    int nxt = cur + (int)gridDim.x; // the first two items are dealt statically, the later ones by the counter
    if (tid == 0) bs->tail_cached = -1;
    __syncthreads();

    float keep[KR][16];
    // the RNA partition of a listed read: signal[min(polya_end, S) : S]
    auto describe = [&](int item, X &xo, int &no) {
        const int ro = list[item];
        const long long S = req[ro].S, pe = req[ro].p_e;
        const long long a = pe < S ? pe : S;
        no = (int)(S - a);
        xo = sigs.row(ro, m) + a;
    };
    // the three pivot samples, then this wave's first KR groups (group w + NW q of the segment).  Every load is issued whatever the
    // length (a group that does not exist reads group w again): straight-line code
    // A wave has at most 63 vector-memory operations in flight (vmcnt): the request comes in two parts -- the pivots and the first RQ1
    // groups before the record is written (its stores must still find room in that count), the other groups behind it.
    constexpr int RQ1 = 3;
    auto request = [&](const X &xo, int no, float (&pv)[3], int part) {
        if (part == 0) { pv[0] = xo.p[no / 4]; pv[1] = xo.p[no / 2]; pv[2] = xo.p[(3 * (long long)no) / 4]; }
        const int ngrp = (no / 8192) * 8;
#pragma unroll
        for (int q = 0; q < KR; q++) {
            if ((q < RQ1) != (part == 0)) continue;
            const int g = w + NW * q;
            rs_load<false>(xo, (long long)(g < ngrp ? g : w) * 1024, ln, keep[q]);
        }
    };

    X x, xn;
    int n, nn;
    float pv[3], pvn[3];
    describe(cur, x, n);
    request(x, n, pv, 0);
    request(x, n, pv, 1);
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
        S1Rec *rec = reinterpret_cast<S1Rec *>(recs + (size_t)cur * S1_REC_FLOATS);
        // ---- pass A: numpy-ordered sum + bucket histogram -------------------------------------------------
        uint32_t wlo;
        {
            const float pivot = fmaxf(fminf(pv[0], pv[1]), fminf(fmaxf(pv[0], pv[1]), pv[2]));
            const uint32_t kb = f2key(pivot) >> BS_KSH;
            wlo = kb >= BS_BINS / 2 ? kb - BS_BINS / 2 : 0u; // centred on the pivot
        }
        for (int i = tid; i < BS_BINS + 4; i += RS_THREADS) bs->hist[i] = 0;
        if (tail > 0) bs_tail_leaves(tail, bs); // (barriers inside)
        __syncthreads();
        RS_PHASE(0);
        RsPassB pb;
        pb.region = sh->special + w * S1_SPECIAL_CAP; pb.dump = sh->dump + ln;
        // two streamed groups in flight per wave, in two register sets used in turn (no copies: a copy would wait for the newer load)
        float pf0[16], pf1[16];
        if (nstream > 0) rs_load<true>(x, gpos(KR), ln, pf0);
        if (nstream > 1) rs_load<true>(x, gpos(KR + 1), ln, pf1);
        RsRagged rg;
        const int nrg = tail > 0 ? (bs->nleaf + 7) >> 3 : 0; // ragged groups: wave w takes w, w + NW, ... (a chunk of more than 64 leaves has 9 .. 16)
        if (w < nrg) rs_ragged_load<true>(xtail, tail, w, ln, bs, rg);
        auto stream_a = [&](float (&c)[16], int i) { // streamed group i of pass A: kept in LDS if it is one of the first KL, summed, its register set refilled
            if (i < KL) {
#pragma unroll
                for (int u = 0; u < 4; u++) { const adp_v4f t4 = {c[4 * u], c[4 * u + 1], c[4 * u + 2], c[4 * u + 3]}; sh->lres[((w * KL + i) * 4 + u) * 64 + ln] = t4; }
            }
            const float s_ = rs_group<0>(c, wlo, pb, bs);
            if (ln == 0) sh->slabsum[w + NW * (KR + i)] = s_;
            if (i + 2 < nstream) rs_load<true>(x, gpos(KR + i + 2), ln, c);
        };
        // a resident group and a streamed one in turn: the streamed groups' loads fly while the resident ones are summed
        static_assert(KR % 2 == 0, "the streamed groups alternate between two register sets: an even count in front of the loop");
#pragma unroll
        for (int q = 0; q < KR; q++) {
            if (q < mine) {
                const float s_ = rs_group<0>(keep[q], wlo, pb, bs);
                if (ln == 0) sh->slabsum[w + NW * q] = s_;
            }
            if (q < nstream) { if (q & 1) stream_a(pf1, q); else stream_a(pf0, q); }
        }
        RS_PHASE(1);
        for (int i = KR; i < nstream; i += 2) {
            stream_a(pf0, i);
            if (i + 1 < nstream) stream_a(pf1, i + 1);
        }
        RS_PHASE(2);
        for (int g = w; g < nrg; g += NW) {
            if (g != w) rs_ragged_load<true>(xtail, tail, g, ln, bs, rg);
            rs_ragged_group<0>(rg, wlo, pb, bs);
        }
        RS_PHASE(3);
        // pass B's first streamed groups (the last ones of pass A) and its first ragged group are requested now: they arrive during the
        // phase between the passes
        const int nsb = nstream > KL ? nstream - KL : 0; // groups pass B streams: KR + KL .. mine - 1, walked from the end
        if (nsb > 0) rs_load<false>(x, gpos(KR + nstream - 1), ln, pf0);
        if (nsb > 1) rs_load<false>(x, gpos(KR + nstream - 2), ln, pf1);
        if (w < nrg) rs_ragged_load<false>(xtail, tail, w, ln, bs, rg);
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
        bool bail = mean != mean; // a NaN, or infinities of both signs: k_partition_stats looks at the segment
        int bin = 0, rk = 0;
        float c = 0.f, w0 = 0.f, P = 0.f, Q = 0.f, hw = 0.f;
        if (!bail) {
            block_find_bin<BS_BINS>(bs, bs_bins(bs), k1, (int)below);
            bail = bs->flag != 0;
            bin = bs->bin; rk = k1 - bs->before;
            __syncthreads();
        }
        if (!bail) {
            const uint32_t key_lo = (wlo + (uint32_t)bin) << BS_KSH;
            const float c_lo = key2f(key_lo), c_hi = key2f(key_lo + (1u << BS_KSH));
            c = 0.5f * (c_lo + c_hi);
            w0 = c_hi - c_lo;
            if (!(w0 > 0.f) || __builtin_isinf(c_hi) || __builtin_isinf(c_lo)) bail = true;
            else {
                const float hm = fmaxf(c - c_lo, c_hi - c) * 1.0001f;
                hw = __uint_as_float(__float_as_uint(hm) + 2u);
            }
        }
        if (!bail) bail = !bs_predict_mad_i(bs, wlo, k1, c, w0, P, Q);
        __syncthreads();
        // ---- pass B: squared deviations + the special samples ------------------------------------------------------
        pb.mean = mean; pb.c = c; pb.P = P; pb.Q = Q; pb.hw = hw; pb.cnt_lt = 0; pb.cur = 0;
        RS_PHASE(4);
        if (!bail) {
            for (int g = w; g < nrg; g += NW) {
                if (g != w) rs_ragged_load<false>(xtail, tail, g, ln, bs, rg);
                rs_ragged_group<1>(rg, wlo, pb, bs);
            }
            RS_PHASE(5);
            auto stream_b = [&](float (&cc)[16], int jb) { // jb-th streamed group of pass B = streamed group nstream - 1 - jb of pass A
                const int i = nstream - 1 - jb;
                const float s_ = rs_group<1>(cc, wlo, pb, bs);
                if (ln == 0) sh->slabsum[w + NW * (KR + i)] = s_;
                if (jb + 2 < nsb) rs_load<false>(x, gpos(KR + i - 2), ln, cc);
            };
            // the groups kept in registers and in LDS, a streamed group behind each
#pragma unroll
            for (int q = 0; q < KR; q++) {
                if (q < mine) {
                    const float s_ = rs_group<1>(keep[q], wlo, pb, bs);
                    if (ln == 0) sh->slabsum[w + NW * q] = s_;
                }
                if (q < nsb) { if (q & 1) stream_b(pf1, q); else stream_b(pf0, q); }
            }
            RS_PHASE(6);
            const int nl = nstream < KL ? nstream : KL;
#pragma unroll
            for (int i = 0; i < KL; i++)
                if (i < nl) {
                    float cc[16];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const adp_v4f t4 = sh->lres[((w * KL + i) * 4 + u) * 64 + ln]; cc[4 * u] = t4.x; cc[4 * u + 1] = t4.y; cc[4 * u + 2] = t4.z; cc[4 * u + 3] = t4.w; }
                    const float s_ = rs_group<1>(cc, wlo, pb, bs);
                    if (ln == 0) sh->slabsum[w + NW * (KR + i)] = s_;
                    if (KR + i < nsb) { if ((KR + i) & 1) stream_b(pf1, KR + i); else stream_b(pf0, KR + i); }
                }
            RS_PHASE(7);
            for (int jb = KR + KL; jb < nsb; jb++) { if (jb & 1) stream_b(pf1, jb); else stream_b(pf0, jb); }
        }
        // ---- the registers are free: the next read's resident groups are requested ---------------------------------------
        RS_PHASE(8);
        unsigned int ticket = 0;
        if (tid == 0) ticket = atomicAdd(work, 1u);
        const bool have_next = nxt < count;
        describe(have_next ? nxt : cur, xn, nn);
        request(xn, nn, pvn, 0);
        RS_PHASE(9);
        // ---- the record: header + the waves' regions packed one behind the other ------------------------------------------
        if (!bail) {
            const uint32_t w2 = (uint32_t)wave_sum((int)pb.cnt_lt);
            if (tid == 0) bs->cntb = 0;
            if (ln == 0) sh->cursor[w] = pb.cur;
            __syncthreads();
            if (ln == 0 && w2) __hip_atomic_fetch_add(&bs->cntb, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int off = 0, total = 0;
            bool over = false;
#pragma unroll
            for (int i = 0; i < NW; i++) { const int ci = sh->cursor[i]; if (i < w) off += ci; total += ci; over |= ci > S1_SPECIAL_CAP; }
            const float sumsq = fold(); // (barriers: the count above is complete behind them)
            bail = over;
            if (!over) {
                float *dst = recs + (size_t)cur * S1_REC_FLOATS + S1_REC_HEAD + off;
                for (int i = ln; i < pb.cur; i += 64) dst[i] = pb.region[i];
                if (tid == 0) {
                    rec->n = n; rec->mean = mean; rec->sumsq = sumsq; rec->wlo = wlo; rec->bin = bin; rec->rk = rk;
                    rec->c = c; rec->w0 = w0; rec->P = P; rec->Q = Q; rec->cnt_lt = bs->cntb; rec->n_special = total;
                }
            }
        }
        if (tid == 0) {
            rec->state = bail ? 2 : 1;
            if (!(g_ablate & 262144)) atomicAdd(&g_bs_tally[blockIdx.x & (ADP_NTALLY - 1)][bail ? 6 : 5], 1ull); // (5: passes done here; 6: left to k_partition_stats)
            sh->item_next = (int)ticket + 2 * (int)gridDim.x;
        }
        request(xn, nn, pvn, 1);
        __syncthreads();
        RS_PHASE(10);
        if (!have_next) break;
        cur = nxt; x = xn; n = nn;
        pv[0] = pvn[0]; pv[1] = pvn[1]; pv[2] = pvn[2];
        nxt = sh->item_next;
        __syncthreads();
    }
}
