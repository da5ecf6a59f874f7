// block_stats.h -- S1 calc_partition_stats (mean, std, median, MAD of a read segment) by one
// 256-thread workgroup per read, in TWO passes over the segment (four when a guess misses):
//
//   pass A  numpy-ordered float32 sum (-> mean)  +  histogram of the top 20 key bits inside a 4096-bucket
//           window around a pivot (0.03 pA buckets near 100 pA): the median's bucket is known, and the
//           histogram also tells, to a few buckets, where the MAD will be;
//   pass B  numpy-ordered sum of (x-mean)^2 (-> std)  +  copy the median's bucket to LDS  +  count the
//           samples whose distance to the bucket centre is below a bracket [P, Q] around the predicted MAD
//           and copy those inside it to LDS.  (Per sample only a subtraction, three compares and a flag bit;
//           the ~2 % flagged samples are re-read and classified exactly.  The largest sample below the
//           median's bucket -- the lower median of an even count that falls on the bucket's first sample --
//           costs a pass of its own in that rare case.)
//   The median is finished exactly inside its bucket; the MAD is finished inside the bracket and
//   ACCEPTED ONLY IF the selected |x - med| values lie at least one bucket width inside [P, Q] -- that
//   proves no sample outside the bracket can sit on the wrong side (|x-med| and |x-centre| differ by
//   less than a bucket width).  Otherwise passes C/D histogram and collect |x - med| as before, and if
//   the window itself missed the generic 4-pass radix select (wave_stats.h) runs: slower, never wrong.
//
// reference: adapted/partition/signal_partitions.py:81-96 (np.mean, np.std, np.median, np.median(|x-med|)
// on float32 slices).  Sums follow numpy's add.reduce association exactly (8192-element chunks in
// sequence; each chunk a balanced tree over 128-element leaves; each leaf 8 interleaved accumulators):
// a half chunk (4096 samples) is staged in LDS with coalesced loads, 256 threads each run one
// accumulator chain of 16 samples, xor-shuffles fold the 8 accumulators and then the 64 leaves.
#pragma once
#include <cstddef>
#include "common.h"
#include "wave_stats.h"

#define BS_THREADS 256
#define BS_LEAF_STRIDE 136
#define BS_BINS 3072      // 20-bit key buckets in the window of pass A (1.5 octaves, centred on the pivot)
#define BS_KSH 12         // key >> 12 = 20-bit bucket index
#define BS_MEDCAP 512     // samples of the median's bucket kept in LDS (behind the MAD bracket's, in the histogram's storage)
#define BS_MADCAP 2560    // samples of the MAD bracket kept in LDS (they reuse the histogram's storage)
#define BS_MAXCHUNK 128   // whole 8192-sample chunks per segment (1 Mi samples); longer segments take the generic path
#define BS_BINS18 1536    // fallback passes C/D: 18-bit buckets, collect capacity 1536

// (defined in adapted_hip.hip) timing experiments only (ADP_ABLATE); results are wrong when non-zero
extern __device__ int g_ablate;
extern __device__ unsigned long long g_bs_tally[ADP_NTALLY][8];
extern __device__ unsigned long long g_dbg[ADP_NDBG]; // debug tallies (adp_debug_fetch what=8): 0-4 here, 5-7 N1, 8-15 phase cycles here

// 16-byte load from a 4-byte aligned address (segments start anywhere): gfx950 global loads need dword alignment only
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// NT_: threads of the workgroup (NW waves, each with eight staging rows); INL_: the selections and bucket searches are inlined
// into the caller (block_stats_res.h: nothing may be called while a read's resident slabs sit in registers)
// STAGE_: floats of the staging area (eight rows per wave for the staged slabs; block_stats_res.h stages nothing and keeps only what
// the bucket search and the selections use as scratch)
template <int NT_, bool INL_ = false, int STAGE_ = (NT_ / 64) * 8 * BS_LEAF_STRIDE>
struct BlockScratchT {
    static constexpr int NT = NT_, NW = NT_ / 64, STAGE = STAGE_;
    static constexpr bool INLINE = INL_;
    union {
        float stage[STAGE_];
        WaveScratch ws; // generic wave-level fallbacks reuse the staging area
    } u;
    uint32_t hist[BS_BINS + 4]; // pass A: below-window cell, BS_BINS bins, above-window cell (bs_bins); pass B: MAD bracket samples (as float); passes C/D: hist18 + collect18
    float chunk_sum[BS_MAXCHUNK]; // sums of the whole numpy chunks of the segment under way
    float tleaf[128]; // leaf sums of a ragged chunk, by tree slot
    int scan[2 * NW];
    int bin, before, ncollect, nmad, flag;
    uint32_t below, cntb;
    uint32_t kmin, kmax; // smallest / largest key copied out of the median's bucket (equal: the bucket holds ONE value)
    float bcast[4];
    int nleaf, tail_cached;
    short leaf_off[132], leaf_len[132], leaf_slot[132]; // numpy's pairwise leaves of a ragged (< 8192) chunk, and their tree slots
    unsigned char slot_used[128];
};
typedef BlockScratchT<BS_THREADS> BlockScratch;

// The thread's index inside a helper.  For the scratch types whose helpers are inlined into ONE long persistent loop (block_stats_res.h)
// it passes through an empty volatile asm: the compiler would otherwise hoist every address and constant a helper derives from the
// index out of that loop and hold them in registers for the whole kernel (65 of 256 there), beside the resident slabs
template <class BS>
static __device__ __forceinline__ int bs_tid()
{
    int t = threadIdx.x;
    if constexpr (BS::INLINE) asm volatile("" : "+v"(t));
    return t;
}

// samples of the median's bucket: behind the MAD bracket's in the histogram's storage (BS_MADCAP + BS_MEDCAP <= BS_BINS)
template <class BS>
static __device__ __forceinline__ LDS float *bs_collect(LDS BS *bs) { return (LDS float *)bs->hist + BS_MADCAP; }
// pass A's histogram: hist[0] counts the samples in buckets BELOW the window, hist[1 .. BS_BINS] are the window's bins, hist[BS_BINS + 1]
// counts those above -- so that a sample's cell is ONE signed clamp of (bucket - window start) to [-1, BS_BINS] (v_med3_i32) plus 1
template <class BS>
static __device__ __forceinline__ LDS uint32_t *bs_bins(LDS BS *bs) { return bs->hist + 1; }
static __device__ __forceinline__ int bs_cell(float v, uint32_t wlo)
{
    const int d = (int)((f2key(v) >> BS_KSH) - wlo); // (20-bit buckets: the difference fits an int; negative below the window)
    const int lo = d < -1 ? -1 : d;
    return lo > BS_BINS ? BS_BINS : lo;
}

static __device__ __forceinline__ float bs_x2(float x, int mode, float c)
{
    if (mode == 0) return x;
    float d = x - c;
    return d * d;
}

// numpy's pairwise recursion over a ragged chunk (< 8192 samples): a node longer than 128 splits into
// (n2, len - n2) with n2 = (len / 2) & ~7; depth <= 7.  Thread t < 128 walks from the root along the bits of t
// (MSB first); a node that is already a leaf stays on the all-zero continuation of its path, so after 7 steps the
// occupied slots, in slot order, are the leaves in numpy's left-to-right order.  The table is kept per tail
// length: the two passes over a segment (and equal tails of later segments) reuse it.
template <class BS>
static __device__ __forceinline__ void bs_tail_leaves(int tail, LDS BS *bs)
{
    const int tid = bs_tid<BS>();
    if (bs->tail_cached == tail) return; // (uniform: written by one thread behind a barrier)
    __syncthreads();
    int off = 0, len = tail;
    bool occupied = tid < 128;
#pragma unroll
    for (int d = 6; d >= 0; d--) {
        const int bit = (tid >> d) & 1;
        if (len <= 128) { if (bit) occupied = false; }
        else {
            int n2 = len / 2;
            n2 -= n2 % 8;
            if (bit) { off += n2; len -= n2; } else len = n2;
        }
    }
    // compact the occupied slots (threads 0..127 = waves 0 and 1) in slot order
    const unsigned long long m = __ballot(occupied);
    if (tid == 0) bs->scan[0] = __popcll(m);
    __syncthreads();
    if (tid < 128) {
        const int idx = (tid >= 64 ? bs->scan[0] : 0) + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (occupied) { bs->leaf_off[idx] = (short)off; bs->leaf_len[idx] = (short)len; bs->leaf_slot[idx] = (short)tid; }
        bs->tleaf[tid] = 0.0f;
        bs->slot_used[tid] = occupied ? 1 : 0;
        if (tid == 64) bs->nleaf = bs->scan[0] + __popcll(m);
    }
    if (tid == 0) bs->tail_cached = tail;
    __syncthreads();
}

// sum of the tree from the leaf sums stored by slot (bs->tleaf, bs->slot_used): seven levels bottom-up; an empty
// right sibling means the left one is carried up unchanged.  Wave 0 works; lane 0 returns the root.
template <class BS>
static __device__ __forceinline__ float bs_tail_tree(LDS BS *bs)
{
    const int ln = bs_tid<BS>(); // (< 64)
    float v0 = bs->tleaf[2 * ln], v1 = bs->tleaf[2 * ln + 1];
    bool u0 = bs->slot_used[2 * ln] != 0, u1 = bs->slot_used[2 * ln + 1] != 0;
    float v = u1 ? v0 + v1 : v0; // level 6 (64 nodes, one per lane)
    bool u = u0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { // levels 5 .. 0: node = lanes aligned to 2*o
        const float w = __shfl_down(v, o);
        const bool wu = __shfl_down((int)u, o) != 0;
        if ((ln & (2 * o - 1)) == 0) { if (wu) v = v + w; }
    }
    return v;
}

enum { SIDE_NONE = 0, SIDE_HIST = 1, SIDE_COLLECT = 2 };
struct SumAux { float sum; uint32_t aux, aux2; };
struct SideParam { uint32_t key; float c, P, Q, hw; int do_mad; }; // hw: |v - c| <= hw for every sample of bucket `key`

// per-sample side effect of a summing pass (state in registers only)
//   SIDE_HIST:    20-bit bucket histogram inside the window starting at p.key; aux counts samples below it
//   SIDE_COLLECT: copy the samples of bucket p.key to LDS; with p.do_mad also classify by dt = |v - p.c|:
//                 dt < P -> aux2++, P <= dt <= Q -> copy to the bracket buffer
template <int SIDE, class BS>
static __device__ __forceinline__ void bs_side(float v, const SideParam &p, LDS BS *bs, uint32_t &aux, uint32_t &aux2)
{
    if (SIDE == SIDE_HIST) {
        __hip_atomic_fetch_add(&bs_bins(bs)[bs_cell(v, p.key)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (SIDE == SIDE_COLLECT) {
        uint32_t key = f2key(v);
        uint32_t kb = key >> BS_KSH;
        if (kb == p.key) {
            int slot = __hip_atomic_fetch_add(&bs->ncollect, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (slot < BS_MEDCAP) bs_collect(bs)[slot] = v;
            __hip_atomic_fetch_min(&bs->kmin, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(&bs->kmax, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (p.do_mad) {
            float dt = fabsf(v - p.c);
            if (dt < p.P) aux2++;
            else if (dt <= p.Q) {
                int slot = __hip_atomic_fetch_add(&bs->nmad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (slot < BS_MADCAP) ((LDS float *)bs->hist)[slot] = v;
            }
        }
    }
}

// The same for four samples at once with the common path free of branches: out-of-window samples of the
// histogram pass go to a dump cell, and the rare copies of the collect pass (about 1.6 % of the samples)
// share ONE divergent branch per four samples.
template <int SIDE, class BS>
static __device__ __forceinline__ void bs_side4(float v0, float v1, float v2, float v3, const SideParam &p, LDS BS *bs,
                                                uint32_t &aux, uint32_t &aux2)
{
    if (SIDE == SIDE_HIST) {
        const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // (the cells on either side of the window's bins take the samples outside it: the count below the window, which the
            // bucket search needs, is read once after the pass instead of a compare and an add per sample)
            __hip_atomic_fetch_add(&bs_bins(bs)[bs_cell(vv[i], p.key)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// the copies of SIDE_COLLECT alone (dt < P is counted elsewhere)
template <class BS>
static __device__ __forceinline__ void bs_copy_exact(float v, const SideParam &p, LDS BS *bs)
{
    const uint32_t key = f2key(v);
    if ((key >> BS_KSH) == p.key) {
        int slot = __hip_atomic_fetch_add(&bs->ncollect, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (slot < BS_MEDCAP) bs_collect(bs)[slot] = v;
        __hip_atomic_fetch_min(&bs->kmin, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(&bs->kmax, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (p.do_mad) {
        const float dt = fabsf(v - p.c);
        if (dt >= p.P && dt <= p.Q) {
            int slot = __hip_atomic_fetch_add(&bs->nmad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (slot < BS_MADCAP) ((LDS float *)bs->hist)[slot] = v;
        }
    }
}

// SIDE_COLLECT for four samples, free of branches: count dt < P and flag (one bit per sample) the candidates for a
// copy -- dt <= hw (a superset of the median's bucket) or P <= dt <= Q.  The flagged samples (about 2 %) are
// re-read and classified exactly by bs_side<SIDE_COLLECT> afterwards.
static __device__ __forceinline__ uint32_t bs_flag4(float v0, float v1, float v2, float v3, const SideParam &p, uint32_t &aux2)
{
    const float vv[4] = {v0, v1, v2, v3};
    uint32_t f = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float dt = fabsf(vv[i] - p.c);
        const bool lt = dt < p.P;
        aux2 += lt ? 1u : 0u;
        const bool special = (dt <= p.hw) || (!lt && dt <= p.Q);
        f |= special ? (1u << i) : 0u;
    }
    return f;
}

// One slab (1024 consecutive samples = eight of numpy's 128-sample leaves) of a summing pass, by one WAVE: lane ln holds samples
// (u * 64 + ln) * 4 .. + 3 in v[u].  The wave stages them in its eight LDS rows, applies the pass's side effect, runs the 8 x 8
// accumulator chains (lane = leaf ln >> 3, accumulator ln & 7, 16 samples each) and folds them by shuffles: the 8 accumulators of
// a leaf, then the slab's 8 leaves (three levels of numpy's balanced tree).  Every lane returns the slab's sum.
template <int SIDE, class BS>
static __device__ __forceinline__ float bs_slab_sum(const float4 (&v)[4], int mode, float c, LDS BS *bs, LDS float *wstage, const SideParam &param,
                                                    uint32_t &aux, uint32_t &aux2)
{
    const int ln = threadIdx.x & 63;
    ws_sync(); // this wave's previous chain reads of its staging rows are done
    uint32_t flags = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int e = (u * 64 + ln) * 4; // four consecutive samples of one leaf
        if (SIDE == SIDE_COLLECT) flags |= bs_flag4(v[u].x, v[u].y, v[u].z, v[u].w, param, aux2) << (4 * u);
        else bs_side4<SIDE>(v[u].x, v[u].y, v[u].z, v[u].w, param, bs, aux, aux2);
        // RAW samples are staged (the transform is applied on the way out): the flagged ones are re-read here
        v4f t4 = {v[u].x, v[u].y, v[u].z, v[u].w};
        *reinterpret_cast<LDS v4f *>(wstage + (e >> 7) * BS_LEAF_STRIDE + (e & 127)) = t4;
    }
    ws_sync();
    if (SIDE == SIDE_COLLECT && flags) { // the few flagged samples: exact classification and copies
        do {
            const int b = __ffs(flags) - 1;
            flags &= flags - 1;
            const int e = ((b >> 2) * 64 + ln) * 4 + (b & 3);
            bs_copy_exact(wstage[(e >> 7) * BS_LEAF_STRIDE + (e & 127)], param, bs);
        } while (flags);
    }
    const LDS float *qq = wstage + (ln >> 3) * BS_LEAF_STRIDE + (ln & 7);
    float r = bs_x2(qq[0], mode, c);
#pragma unroll
    for (int t = 1; t < 16; t++) r += bs_x2(qq[8 * t], mode, c);
    r = r + __shfl_xor(r, 1);   // the 8 accumulators of a leaf
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    r = r + __shfl_xor(r, 8);   // the slab's 8 leaves: three levels of numpy's balanced tree
    r = r + __shfl_xor(r, 16);
    r = r + __shfl_xor(r, 32);
    return r;
}

// numpy-ordered sum of xf(xt[0..tail)), 0 < tail < 8192 (the ragged chunk behind a segment's whole chunks), with the per-sample side
// effect riding on its loads; lane 0 of wave 0 returns the sum (the other threads 0).  All threads of the workgroup call it.
template <int SIDE, class X, class BS>
static __device__ __forceinline__ float bs_ragged_sum(X xt, int tail, int mode, float c, LDS BS *bs, const SideParam &param, uint32_t &aux,
                                                      uint32_t &aux2)
{
    constexpr int NW = BS::NW;
    const int tid = threadIdx.x;
    // leaves of numpy's pairwise recursion over the ragged chunk (split n -> n2 = (n/2) & ~7, n - n2)
    bs_tail_leaves(tail, bs);
    const int nleaf = bs->nleaf;
    for (int g0 = 0; g0 < nleaf; g0 += 8 * NW) {
        // stage 8 NW leaves: wave w loads leaves g0 + 8w .. g0 + 8w + 7 (coalesced, 2 loads per leaf)
        __syncthreads();
        const int w = tid >> 6, ln = tid & 63;
        // (all sixteen loads of the wave's eight leaves first -- clamped indices, no conditions --, then the LDS writes:
        // load, wait, write per leaf cost sixteen memory round trips per group of 32 leaves)
        float va[8], vb[8];
        int lens[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int l = g0 + w * 8 + q;
            const bool have = l < nleaf;
            const int off = have ? bs->leaf_off[l] : 0, len = have ? bs->leaf_len[l] : 0;
            const int last = len > 0 ? len - 1 : 0;
            lens[q] = len;
            va[q] = xt[off + (ln < last ? ln : last)];
            vb[q] = xt[off + (ln + 64 < last ? ln + 64 : last)];
        }
        // (the side effects of the ragged part ride on these loads: the leaves cover every sample of it exactly once)
#pragma unroll
        for (int q = 0; q < 8; q++) {
            LDS float *dst = bs->u.stage + (w * 8 + q) * BS_LEAF_STRIDE;
            if (ln < lens[q]) { dst[ln] = bs_x2(va[q], mode, c); if (SIDE != SIDE_NONE) bs_side<SIDE>(va[q], param, bs, aux, aux2); }
            if (ln + 64 < lens[q]) { dst[ln + 64] = bs_x2(vb[q], mode, c); if (SIDE != SIDE_NONE) bs_side<SIDE>(vb[q], param, bs, aux, aux2); }
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
            bs->tleaf[bs->leaf_slot[l]] = res;
        }
    }
    __syncthreads();
    return tid < 64 ? bs_tail_tree(bs) : 0.0f; // (lane 0 holds the root)
}

// numpy-ordered sum of xf(x[0..n)) fused with a per-sample side effect; all threads return the sum and the
// block-reduced aux (SIDE_HIST: count below the window; SIDE_COLLECT: max key below the bucket, aux2 = count
// of samples closer to the centre than the bracket)
template <int SIDE, class X, class BS>
static __device__ __noinline__ SumAux block_np_sum(X x, int n, int mode, float c, LDS BS *bs,
                                                   SideParam param)
{
    const int tid = threadIdx.x;
    uint32_t aux = 0, aux2 = 0;
    float total = 0.0f; // meaningful in wave 0
    int s = 0;
#ifdef ADP_PHASE_TIMING
    long long tph = clock64();
    auto phase = [&](int slot) { // (debug build) 25 + 4 * SIDE: whole chunks, then the ragged part's side effects, its leaves and tree, the epilogue
        long long t = clock64();
        if (tid == 0 && n >= 8192) atomicAdd(&g_dbg[slot + 4 * SIDE], (unsigned long long)(t - tph));
        tph = t;
    };
#else
    auto phase = [](int) {};
#endif
    // Each WAVE owns whole numpy chunks (8192 samples; wave w takes chunks w, w+NW, ...): it streams the chunk in eight
    // slabs of 1024 samples (8 leaves), stages each slab in its own LDS rows, runs the 8 x 8 accumulator chains, folds
    // them by shuffles and keeps the slab sums in lanes 0..7; a last butterfly over those lanes is the top of numpy's
    // balanced tree.  No block-wide barrier inside the stream: the chunk sums meet once, in order, at the end.
    constexpr int NW = BS::NW;
    const int w = tid >> 6, ln = tid & 63;
    LDS float *wstage = bs->u.stage + w * 8 * BS_LEAF_STRIDE;
    const int nchunk = n / 8192;
    // fewer chunks than waves (a short RNA part): the SLABS go round the waves instead -- wave w takes slabs w, w + 4, ... of the
    // 8 * nchunk -- so that all four stream; the slab sums then meet in LDS (bs->tleaf, free until the ragged part) for the chunks' trees
    const bool slabwise = nchunk <= 16; // (tleaf holds 128 slab sums)
    const int myslabs = slabwise ? (8 * nchunk - w + NW - 1) / NW : (nchunk > w ? ((nchunk - w + NW - 1) / NW) * 8 : 0); // slabs this wave streams
    auto slab_off = [&](int q) { return slabwise ? (long long)(w + NW * q) * 1024 : (long long)(w + NW * (q >> 3)) * 8192 + (q & 7) * 1024; }; // first sample of slab q
    // software pipeline: the loads of the next PF slabs fly (as raw samples) while this one is summed
    constexpr int PF = X::PREFETCH;
    typename X::Raw4 pf[PF][4];
#pragma unroll
    for (int d = 0; d < PF; d++)
        if (d < myslabs) {
            const long long p = slab_off(d);
#pragma unroll
            for (int u = 0; u < 4; u++) pf[d][u] = x.raw4u_in(p + (u * 64 + ln) * 4);
        }
    float slabsum = 0.0f; // lane j (< 8): sum of slab j of the current chunk
    for (int q = 0; q < myslabs; q++) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = x.cook4(pf[0][u]);
#pragma unroll
        for (int d = 0; d + 1 < PF; d++)
#pragma unroll
            for (int u = 0; u < 4; u++) pf[d][u] = pf[d + 1][u];
        if (q + PF < myslabs) {
            const long long p = slab_off(q + PF);
#pragma unroll
            for (int u = 0; u < 4; u++) pf[PF - 1][u] = x.raw4u_in(p + (u * 64 + ln) * 4);
        }
        const float r = bs_slab_sum<SIDE>(v, mode, c, bs, wstage, param, aux, aux2);
        if (slabwise) { if (ln == 0) bs->tleaf[w + NW * q] = r; continue; }
        if (ln == (q & 7)) slabsum = r;
        if ((q & 7) == 7) { // chunk complete: the top three levels over its 8 slab sums (lanes 0..7)
            float cs = slabsum;
            cs = cs + __shfl_xor(cs, 1);
            cs = cs + __shfl_xor(cs, 2);
            cs = cs + __shfl_xor(cs, 4);
            const int ch = w + NW * (q >> 3);
            if (ln == 0) bs->chunk_sum[ch & (BS_MAXCHUNK - 1)] = cs;
        }
    }
    __syncthreads();
    if (slabwise && nchunk > 0) {
        if (tid < nchunk) {
            const LDS float *t = bs->tleaf + 8 * tid;
            bs->chunk_sum[tid] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        __syncthreads();
    }
    if (tid == 0) for (int ch = 0; ch < nchunk; ch++) total += bs->chunk_sum[ch]; // numpy adds the chunk sums in sequence
    s = nchunk * 8192;
    __syncthreads();
    phase(25);
    const int tail = n - s;
    if (tail > 0 && !(g_ablate & 4)) {
        phase(26);
        const float rt = bs_ragged_sum<SIDE>(x + s, tail, mode, c, bs, param, aux, aux2);
        if (tid < 64) total += rt;
        phase(27);
    }
    __syncthreads();
    if (tid == 0) { bs->bcast[0] = total; bs->below = 0; bs->cntb = 0; }
    __syncthreads();
    if (SIDE == SIDE_HIST) {
        if (tid == 0) bs->below = bs->hist[0]; // samples in buckets below the window (the cell in front of the bins)
    } else if (SIDE == SIDE_COLLECT) {
        uint32_t w = wave_max(aux);
        if ((tid & 63) == 0 && w) __hip_atomic_fetch_max(&bs->below, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t w2 = (uint32_t)wave_sum((int)aux2);
        if ((tid & 63) == 0 && w2) __hip_atomic_fetch_add(&bs->cntb, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    SumAux r;
    r.sum = bs->bcast[0];
    r.aux = bs->below;
    r.aux2 = bs->cntb;
    __syncthreads();
    phase(28);
    return r;
}

// x_(k) and x_(k-1) of xform(x[0..n)) for an LDS array of at most BSEL_CAP elements, by ALL 256 threads (the semantics of
// wave_select2_impl: NaN if any element is NaN; MSB-first radix select over key - min, 8 bits of the span per pass).  A thread
// keeps its <= 12 keys in registers; a pass is one histogram of LDS atomics and two barriers (two histograms in turn), every
// wave locates the bin for itself.  One wave doing the same on its own (wave_select2_lds) took 40-46 k cycles for the 1470 samples
// of a poly(A) slice and 30-60 k for the few dozen of a median bucket / MAD bracket: a quarter of this kernel's time at short windows.
// Scratch: the head of the staging rows (x lives in the histogram's storage).  All threads return the values.
#define BSEL_CAP 3072
template <class BS>
static __device__ __forceinline__ void block_select2_lds_i(const LDS float *x, int n, int k, int mode, float c, LDS BS *bs,
                                                         float &vk, float &vkm1)
{
    const int tid = bs_tid<BS>(), ln = tid & 63;
    LDS uint32_t *sc = (LDS uint32_t *)bs->u.stage; // [0, 512): two histograms; 512: min key, 513: max key, 514: below, 515: NaN seen
    uint32_t key[BSEL_CAP / BS::NT];
    uint32_t mn = 0xffffffffu, mx = 0u;
    bool has_nan = false;
#pragma unroll
    for (int u = 0; u < BSEL_CAP / BS::NT; u++) {
        const int i = tid + u * BS::NT;
        key[u] = 0u;
        if (i < n) {
            const float xf = ws_xform(x[i], mode, c);
            has_nan |= xf != xf;
            const uint32_t kk = f2key(xf);
            key[u] = kk; mn = kk < mn ? kk : mn; mx = kk > mx ? kk : mx;
        }
    }
    if (tid < 4) sc[512 + tid] = tid == 0 ? 0xffffffffu : 0u;
    __syncthreads();
    mn = wave_min(mn); mx = wave_max(mx);
    const bool wnan = __any(has_nan);
    if (ln == 0) {
        __hip_atomic_fetch_min(&sc[512], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(&sc[513], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (wnan) sc[515] = 1u;
    }
    __syncthreads();
    mn = sc[512]; mx = sc[513];
    if (sc[515]) { vk = __builtin_nanf(""); vkm1 = vk; __syncthreads(); return; }
    const uint32_t span = mx - mn;
    int rb = span ? 32 - __clz(span) : 0; // bits of d = key - min still unresolved
    uint32_t prefix = 0, krem = (uint32_t)k, below = 0;
    int lowbin = -1, lastw = 0, par = 0;
    while (rb > 0) {
        const int w = rb < 8 ? rb : 8;
        const int shift = rb - w;
        LDS uint32_t *hist = sc + par * 256;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < BSEL_CAP / BS::NT; u++) {
            const int i = tid + u * BS::NT;
            const uint32_t d = key[u] - mn;
            const uint32_t top = (rb >= 32) ? 0u : (d >> rb);
            if (i < n) {
                if (top == prefix) __hip_atomic_fetch_add(&hist[(d >> shift) & ((1u << w) - 1u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (shift == 0 && top < prefix && d + 1u > below) below = d + 1u;
            }
        }
        __syncthreads();
        // the bin of rank krem: each lane owns 4 consecutive bins (every wave for itself: no third barrier)
        const uint32_t h0 = hist[4 * ln], h1 = hist[4 * ln + 1], h2 = hist[4 * ln + 2], h3 = hist[4 * ln + 3];
        const int s = (int)(h0 + h1 + h2 + h3);
        const int incl = wave_scan_incl(s);
        const int excl = incl - s;
        const bool mine = (int)krem >= excl && (int)krem < incl;
        int bin = 0, before = 0;
        if (mine) {
            int cacc = excl;
            if ((int)krem < cacc + (int)h0) { bin = 4 * ln; before = cacc; }
            else { cacc += h0;
                if ((int)krem < cacc + (int)h1) { bin = 4 * ln + 1; before = cacc; }
                else { cacc += h1;
                    if ((int)krem < cacc + (int)h2) { bin = 4 * ln + 2; before = cacc; }
                    else { cacc += h2; bin = 4 * ln + 3; before = cacc; } } }
        }
        const unsigned long long mm = __ballot(mine);
        const int src = __ffsll((long long)mm) - 1;
        bin = __shfl(bin, src);
        before = __shfl(before, src);
        if (shift == 0) {
            int cand = -1; // largest non-empty bin below `bin` among this lane's four
            if (4 * ln < bin && h0) cand = 4 * ln;
            if (4 * ln + 1 < bin && h1) cand = 4 * ln + 1;
            if (4 * ln + 2 < bin && h2) cand = 4 * ln + 2;
            if (4 * ln + 3 < bin && h3) cand = 4 * ln + 3;
            lowbin = wave_max(cand);
            lastw = w;
        }
        prefix = (prefix << w) | (uint32_t)bin;
        krem -= (uint32_t)before;
        rb = shift;
        par ^= 1;
    }
    vk = key2f(mn + prefix);
    vkm1 = vk;
    if (krem == 0 && k > 0) { // first of its key: the value before it is the largest key below
        below = wave_max(below);
        if (ln == 0 && below) __hip_atomic_fetch_max(&sc[514], below, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        below = sc[514];
        uint32_t d0 = below ? below - 1u : 0u;
        bool have = below != 0;
        if (span && lowbin >= 0) {
            uint32_t da = (prefix & ~((1u << lastw) - 1u)) | (uint32_t)lowbin;
            if (!have || da > d0) d0 = da;
            have = true;
        }
        if (have) vkm1 = key2f(mn + d0);
    }
    __syncthreads();
}

template <class BS>
static __device__ __noinline__ void block_select2_lds_call(const LDS float *x, int n, int k, int mode, float c, LDS BS *bs, float &vk, float &vkm1)
{
    block_select2_lds_i(x, n, k, mode, c, bs, vk, vkm1);
}
// (a called function by default; inlined for the scratch types that ask for it: BlockScratchT<.., true>)
template <class BS>
static __device__ __forceinline__ void block_select2_lds(const LDS float *x, int n, int k, int mode, float c, LDS BS *bs, float &vk, float &vkm1)
{
    if constexpr (BS::INLINE) block_select2_lds_i(x, n, k, mode, c, bs, vk, vkm1);
    else block_select2_lds_call(x, n, k, mode, c, bs, vk, vkm1);
}

// A SHORT segment (the poly(A) slice, the adapter) is read from global memory ONCE, into LDS; its sums and selections run there.
// BS_SMALLCAP: the staging rows and the histogram storage taken together (they are adjacent in BlockScratch).
#define BS_SMALLCAP (BS::STAGE + BS_BINS)
static_assert(offsetof(BlockScratch, hist) == offsetof(BlockScratch, u) + 32 * BS_LEAF_STRIDE * 4, "the histogram's storage must follow the staging rows");
template <int NT, class X>
static __device__ __forceinline__ void bs_copy_to_lds(X x, int n, LDS float *cp)
{
    const int tid = threadIdx.x;
    for (int base = 0; base < n; base += NT * 8) { // eight loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + u * NT + tid; v[u] = ld_if(x, i, i < n); }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + u * NT + tid; if (i < n) cp[i] = v[u]; }
    }
}
// numpy-ordered sum of xf(cp[0..n)), n < 8192 (one ragged numpy chunk: the leaves of bs_tail_leaves, 8 accumulators each, the
// balanced tree above them -- exactly the ragged part of block_np_sum, read from LDS instead of staged from global memory).
// All threads return the sum.  The caller has a barrier between its writes of cp and this call.
template <class BS>
static __device__ __noinline__ float bs_lds_np_sum(const LDS float *cp, int n, int mode, float c, LDS BS *bs)
{
    const int tid = threadIdx.x;
    bs_tail_leaves(n, bs);
    const int nleaf = bs->nleaf;
    for (int g0 = 0; g0 < nleaf; g0 += BS::NT / 8) {
        const int ll = tid >> 3, j = tid & 7; // accumulator chain j of leaf g0 + ll
        const int l = g0 + ll;
        const bool have = l < nleaf;
        const int len = have ? bs->leaf_len[l] : 0;
        const LDS float *q = cp + (have ? bs->leaf_off[l] : 0);
        float r = 0.0f;
        if (len >= 8) {
            r = bs_x2(q[j], mode, c);
            const int lim = len - (len % 8);
            for (int i = 8; i < lim; i += 8) r += bs_x2(q[i + j], mode, c);
        }
        r = r + __shfl_xor(r, 1);
        r = r + __shfl_xor(r, 2);
        r = r + __shfl_xor(r, 4);
        if (j == 0 && have) {
            float res;
            if (len >= 8) { res = r; for (int i = len - (len % 8); i < len; i++) res += bs_x2(q[i], mode, c); }
            else { res = 0.0f; for (int i = 0; i < len; i++) res += bs_x2(q[i], mode, c); }
            bs->tleaf[bs->leaf_slot[l]] = res;
        }
    }
    __syncthreads();
    float total = 0.0f;
    if (tid < 64) total = bs_tail_tree(bs); // (lane 0 holds the root)
    if (tid == 0) bs->bcast[0] = total;
    __syncthreads();
    const float out = bs->bcast[0];
    __syncthreads();
    return out;
}

// window start (in buckets of 2^SH keys): half an octave below the pivot's octave
template <int SH>
static __device__ __forceinline__ uint32_t bs_window_lo(float pivot)
{
    const uint32_t per_oct = 1u << (23 - SH);
    uint32_t kb = f2key(pivot) >> SH;
    uint32_t oct = kb & ~(per_oct - 1u);
    return oct >= per_oct / 2 ? oct - per_oct / 2 : 0u;
}

// locate the bucket holding rank k in h[0..NB) (with `under` samples before it); sets bs->bin/before,
// bs->flag = 1 if the rank lies outside
template <int NB, class BS>
static __device__ __forceinline__ void block_find_bin_i(LDS BS *bs, const LDS uint32_t *hh, int k, int under)
{
    const int tid = bs_tid<BS>();
    constexpr int PER = (NB + BS::NT - 1) / BS::NT;
    uint32_t h[PER];
    int s = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { h[j] = (NB % BS::NT == 0 || tid * PER + j < NB) ? hh[tid * PER + j] : 0u; s += (int)h[j]; }
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
        for (int j = 0; j < PER; j++) {
            if (k >= cacc && k < cacc + (int)h[j]) { bs->bin = tid * PER + j; bs->before = cacc; bs->flag = 0; }
            cacc += (int)h[j];
        }
    }
    __syncthreads();
}

template <int NB, class BS>
static __device__ __noinline__ void block_find_bin_call(LDS BS *bs, const LDS uint32_t *hh, int k, int under) { block_find_bin_i<NB>(bs, hh, k, under); }
template <int NB, class BS>
static __device__ __forceinline__ void block_find_bin(LDS BS *bs, const LDS uint32_t *hh, int k, int under)
{
    if constexpr (BS::INLINE) block_find_bin_i<NB>(bs, hh, k, under);
    else block_find_bin_call<NB>(bs, hh, k, under);
}

struct SegStats { float mean, sd, med, mad; };

// exact median from the collected bucket (wave 0), given the largest key below the bucket
template <class BS>
static __device__ __forceinline__ float bs_median_from_bucket_i(LDS BS *bs, const LDS float *buf, int cnt, int n, int rk,
                                                              uint32_t below_key)
{
    __syncthreads();
    float vk, vkm1;
    block_select2_lds(buf, cnt, rk, 0, 0.f, bs, vk, vkm1); // (cnt <= BS_MEDCAP)
    float res = vk;
    if ((n & 1) == 0) {
        float lo = (rk >= 1) ? vkm1 : key2f(below_key);
        res = (lo + vk) / 2.0f;
    }
    return res;
}

template <class BS>
static __device__ __noinline__ float bs_median_from_bucket(LDS BS *bs, const LDS float *buf, int cnt, int n, int rk, uint32_t below_key)
{
    return bs_median_from_bucket_i(bs, buf, cnt, n, rk, below_key);
}

// largest key below `key_lo` among x[0..n) (0 if none) -- all threads return it
template <class X, class BS>
static __device__ __noinline__ uint32_t bs_max_key_below(X x, int n, uint32_t key_lo, LDS BS *bs)
{
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) bs->below = 0;
    __syncthreads();
    uint32_t best = 0;
    for (int i = tid; i < n; i += BS::NT) { uint32_t key = f2key(x[i]); if (key < key_lo && key > best) best = key; }
    best = wave_max(best);
    if ((tid & 63) == 0 && best) __hip_atomic_fetch_max(&bs->below, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    best = bs->below;
    __syncthreads();
    return best;
}

// The median's bucket holds more samples than the copy list (BS_MEDCAP): data with few distinct values, e.g. calibrated
// int16 ADC samples (a step of ~0.18 pA is wider than a bucket, so thousands of samples share one key).  One more pass
// counts the bucket's samples by their exact key -- 4096 counters in the staging area -- and notes the largest key below
// the bucket; the median follows from the counts, whatever the multiplicities.
template <class X, class BS>
static __device__ __noinline__ float bs_median_dense_bucket(X x, int n, LDS BS *bs, uint32_t key_lo, int rk)
{
    const int tid = threadIdx.x;
    LDS uint32_t *sub = (LDS uint32_t *)bs->u.stage; // 4096 counters (16 KB of the 17 KB staging area)
    static_assert(sizeof(((BS *)0)->u.stage) >= 4096 * 4, "staging area too small for the key counters");
    __syncthreads();
    for (int i = tid; i < 4096; i += BS::NT) sub[i] = 0;
    if (tid == 0) bs->below = 0;
    __syncthreads();
    uint32_t best = 0;
    for (int base = 0; base < n; base += BS::NT * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { int i = base + u * BS::NT + tid; v[u] = ld_if(x, i, i < n); } // (entries beyond n are not used)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = base + u * BS::NT + tid;
            if (i < n) {
                const uint32_t key = f2key(v[u]);
                const uint32_t d = key - key_lo; // wraps below the bucket
                if (d < 4096u) __hip_atomic_fetch_add(&sub[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (key < key_lo && key > best) best = key;
            }
        }
    }
    best = wave_max(best);
    if ((tid & 63) == 0 && best) __hip_atomic_fetch_max(&bs->below, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    block_find_bin<4096>(bs, sub, rk, 0);
    const int bin = bs->bin, rkk = rk - bs->before; // (the rank lies inside: pass A counted the same samples)
    __syncthreads();
    // x_(k-1): the same key when the rank is not the first of its key, else the nearest occupied key below
    if (tid == 0) bs->cntb = 0; // (largest occupied counter index below `bin`, + 1)
    __syncthreads();
    uint32_t lowb = 0;
    for (int i = tid; i < bin; i += BS::NT) if (sub[i]) lowb = (uint32_t)i + 1u;
    lowb = wave_max(lowb);
    if ((tid & 63) == 0 && lowb) __hip_atomic_fetch_max(&bs->cntb, lowb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const float vk = key2f(key_lo + (uint32_t)bin);
    float res = vk;
    if ((n & 1) == 0) {
        float lo = vk;
        if (rkk == 0) lo = bs->cntb ? key2f(key_lo + bs->cntb - 1u) : key2f(bs->below);
        res = (lo + vk) / 2.0f;
    }
    __syncthreads();
    return res;
}

// passes C and D: exact median of |x - med| by an 18-bit window histogram + bucket collection
template <class X, class BS>
static __device__ __noinline__ float bs_mad_two_pass(X x, int n, LDS BS *bs, float med, float sd)
{
    const int tid = threadIdx.x;
    const int k1 = n / 2;
    LDS uint32_t *h18 = bs->hist;
    LDS float *c18 = (LDS float *)(bs->hist + BS_BINS18);
    float pivot = 0.6745f * sd;
    if (!(pivot > 0.f)) pivot = 1.0f;
    const uint32_t wlo = bs_window_lo<14>(pivot) >= 512u ? bs_window_lo<14>(pivot) - 512u : 0u; // [pivot_oct/4, pivot_oct*4)
    __syncthreads();
    for (int i = tid; i < BS_BINS18; i += BS::NT) h18[i] = 0;
    __syncthreads();
    int under = 0;
    for (int base = 0; base < n; base += BS::NT * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { int i = base + u * BS::NT + tid; v[u] = ld_if(x, i, i < n); } // (entries beyond n are not used)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int i = base + u * BS::NT + tid;
            if (i < n) {
                uint32_t k18 = f2key(fabsf(v[u] - med)) >> 14;
                if (k18 < wlo) under++;
                else if (k18 - wlo < (uint32_t)BS_BINS18) __hip_atomic_fetch_add(&h18[k18 - wlo], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    under = wave_sum(under);
    __syncthreads();
    if ((tid & 63) == 0) bs->scan[BS::NW + (tid >> 6)] = under;
    __syncthreads();
    under = 0;
    for (int w_ = 0; w_ < BS::NW; w_++) under += bs->scan[BS::NW + w_];
    block_find_bin<BS_BINS18>(bs, h18, k1, under);
    const bool miss = bs->flag != 0;
    const int bin = bs->bin, rk = k1 - bs->before;
    __syncthreads();
    if (tid == 0) { bs->ncollect = 0; bs->below = 0; bs->kmin = 0xffffffffu; bs->kmax = 0u; }
    __syncthreads();
    if (!miss) {
        const uint32_t tgt = wlo + (uint32_t)bin;
        uint32_t below = 0;
        for (int base = 0; base < n; base += BS::NT * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { int i = base + u * BS::NT + tid; v[u] = ld_if(x, i, i < n); } // (entries beyond n are not used)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int i = base + u * BS::NT + tid;
                if (i < n) {
                    float d = fabsf(v[u] - med);
                    uint32_t key = f2key(d);
                    uint32_t k18 = key >> 14;
                    if (k18 == tgt) {
                        int slot = __hip_atomic_fetch_add(&bs->ncollect, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (slot < BS_BINS18) c18[slot] = d;
                        __hip_atomic_fetch_min(&bs->kmin, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_max(&bs->kmax, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    else if (k18 < tgt && key > below) below = key;
                }
            }
        }
        below = wave_max(below);
        if ((tid & 63) == 0 && below) __hip_atomic_fetch_max(&bs->below, below, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
    }
    float mad;
    if (!miss && bs->ncollect > BS_BINS18 && bs->kmin == bs->kmax) {
        // the bucket of the rank holds one distance only (quantised data): nothing to select
        const float vk = key2f(bs->kmin);
        mad = vk;
        if ((n & 1) == 0) mad = (((rk >= 1) ? vk : key2f(bs->below)) + vk) / 2.0f;
        __syncthreads();
    } else if (miss || bs->ncollect > BS_BINS18) {
        __syncthreads();
        if (tid < 64) { float m_ = wave_median(x, n, 1, med, &bs->u.ws); if (tid == 0) bs->bcast[1] = m_; }
        __syncthreads();
        mad = bs->bcast[1];
        __syncthreads();
    } else {
        mad = bs_median_from_bucket(bs, c18, bs->ncollect, n, rk, bs->below);
    }
    return mad;
}

// After pass A: predict where the MAD lies from the bucket histogram.  Returns false if no prediction.
// c = centre of the median's bucket, w0 = its width; [P, Q] = bracket of distances to c.
template <class BS>
static __device__ __forceinline__ bool bs_predict_mad_i(LDS BS *bs, uint32_t wlo, int k1, float c, float w0, float &P, float &Q)
{
    const int tid = bs_tid<BS>();
    LDS uint32_t *dh = (LDS uint32_t *)bs->u.stage; // distance histogram, BS_BINS cells of width w0
    __syncthreads();
    for (int i = tid; i < BS_BINS; i += BS::NT) dh[i] = 0;
    __syncthreads();
    for (int i = tid; i < BS_BINS; i += BS::NT) {
        uint32_t h = bs_bins(bs)[i];
        if (h) {
            float xc = key2f(((wlo + (uint32_t)i) << BS_KSH) + (1u << (BS_KSH - 1)));
            float d = fabsf(xc - c) / w0;
            if (d < (float)BS_BINS) __hip_atomic_fetch_add(&dh[(int)d], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    block_find_bin<BS_BINS>(bs, dh, k1, 0);
    const bool ok = bs->flag == 0;
    const int j = bs->bin;
    __syncthreads();
    if (!ok) return false;
    P = (float)(j - 3) * w0;
    if (P < 0.f) P = 0.f;
    Q = (float)(j + 4) * w0;
    return true;
}

template <class BS>
static __device__ __noinline__ bool bs_predict_mad(LDS BS *bs, uint32_t wlo, int k1, float c, float w0, float &P, float &Q)
{
    return bs_predict_mad_i(bs, wlo, k1, c, w0, P, Q);
}

// mean / std / median / MAD of x[0..n), n >= 1.  have_medmad: reuse med_in / mad_in (adapter partition).
template <class X, class BS>
static __device__ SegStats block_segment_stats(X x, int n, LDS BS *bs, bool have_medmad,
                                               float med_in, float mad_in)
{
    const int tid = threadIdx.x;
    SegStats o;
    const int k1 = n / 2;
#ifdef ADP_PHASE_TIMING
    long long tph = clock64();
    auto phase = [&](int slot) { // (debug build) shader cycles per phase, summed over the blocks' first threads
        long long t = clock64();
        if (tid == 0) atomicAdd(&g_dbg[8 + slot], (unsigned long long)(t - tph));
        tph = t;
    };
#else
    auto phase = [](int) {};
#endif
    SideParam sp; sp.key = 0; sp.c = 0.f; sp.P = 0.f; sp.Q = 0.f; sp.do_mad = 0;
    if (!have_medmad && n <= BS_BINS) {
        // a segment that fits the histogram's storage (the poly(A) slice): ONE read into LDS, the two numpy-ordered sums and the
        // exact median and MAD by direct selection there -- none of the bucket / bracket machinery
#ifdef ADP_PHASE_TIMING
        long long tq = clock64();
        auto sub = [&](int slot) { long long t = clock64(); if (tid == 0) atomicAdd(&g_dbg[slot], (unsigned long long)(t - tq)); tq = t; };
#else
        auto sub = [](int) {};
#endif
        LDS float *cp = (LDS float *)bs->hist;
        __syncthreads();
        bs_copy_to_lds<BS::NT>(x, n, cp);
        __syncthreads();
        sub(42);
        o.mean = bs_lds_np_sum(cp, n, 0, 0.f, bs) / (float)n;
        sub(40);
        o.sd = sqrtf(bs_lds_np_sum(cp, n, 2, o.mean, bs) / (float)n);
        sub(41);
        float vk, vkm1;
        block_select2_lds(cp, n, k1, 0, 0.f, bs, vk, vkm1);
        o.med = (n & 1) ? vk : (vkm1 + vk) / 2.0f;
        sub(43);
        block_select2_lds(cp, n, k1, 1, o.med, bs, vk, vkm1);
        o.mad = (n & 1) ? vk : (vkm1 + vk) / 2.0f;
        sub(44);
        phase(16);
        return o;
    }
    if (have_medmad && n <= BS_SMALLCAP) {
        // the adapter (its median and MAD come from k_validate): one read into LDS, the two sums there
        LDS float *cp = (LDS float *)bs->u.stage; // (runs on into the histogram's storage)
        __syncthreads();
        bs_copy_to_lds<BS::NT>(x, n, cp);
        __syncthreads();
        o.mean = bs_lds_np_sum(cp, n, 0, 0.f, bs) / (float)n;
        if (o.mean != o.mean) { // (a NaN or infinities of both signs: np.std is NaN either way)
            o.sd = __builtin_nanf("");
        } else o.sd = sqrtf(bs_lds_np_sum(cp, n, 2, o.mean, bs) / (float)n);
        o.med = med_in; o.mad = mad_in;
        return o;
    }
    // ---- pass A: mean + bucket histogram --------------------------------------------------
    uint32_t wlo = 0;
    if (!have_medmad) {
        float a = x[n / 4], b = x[n / 2], c3 = x[(3 * (long long)n) / 4];
        float pivot = fmaxf(fminf(a, b), fminf(fmaxf(a, b), c3));
        { const uint32_t kb = f2key(pivot) >> BS_KSH; wlo = kb >= BS_BINS / 2 ? kb - BS_BINS / 2 : 0u; } // centred on the pivot
    }
    __syncthreads();
    for (int i = tid; i < BS_BINS + 4; i += BS::NT) bs->hist[i] = 0; // (the bins and the dump cells behind them)
    __syncthreads();
    sp.key = wlo;
    SumAux p1 = (have_medmad || (g_ablate & 2048)) ? block_np_sum<SIDE_NONE>(x, n, 0, 0.f, bs, sp) : block_np_sum<SIDE_HIST>(x, n, 0, 0.f, bs, sp);
    o.mean = p1.sum / (float)n;
    if (o.mean != o.mean) {
        // a NaN mean: a NaN sample (np.std, np.median and the MAD are NaN as well then) or infinities of both signs (they are
        // not).  Never the case for the reads of a run; one look at the segment decides.
        __syncthreads();
        if (tid == 0) bs->flag = 0;
        __syncthreads();
        bool nanhere = false;
        for (int i = tid; i < n; i += BS::NT) { const float v = x[i]; nanhere |= v != v; }
        if (__any(nanhere) && (tid & 63) == 0) bs->flag = 1;
        __syncthreads();
        const bool any_nan = bs->flag != 0;
        __syncthreads();
        if (any_nan) { o.sd = o.med = o.mad = __builtin_nanf(""); return o; }
    }
    phase(0);
    bool fallback_med = false, predicted = false;
    int bin = 0, rk = 0;
    float c = 0.f, w0 = 0.f, P = 0.f, Q = 0.f, hw = 0.f;
    if (!have_medmad) {
        block_find_bin<BS_BINS>(bs, bs_bins(bs), k1, (int)p1.aux);
        fallback_med = bs->flag != 0;
        bin = bs->bin; rk = k1 - bs->before;
        __syncthreads();
        if (!fallback_med) {
            // centre and (conservative) half width of the median's bucket
            const uint32_t key_lo = (wlo + (uint32_t)bin) << BS_KSH;
            const float c_lo = key2f(key_lo), c_hi = key2f(key_lo + (1u << BS_KSH));
            c = 0.5f * (c_lo + c_hi);
            w0 = c_hi - c_lo;
            if (!(w0 > 0.f) || __builtin_isinf(c_hi) || __builtin_isinf(c_lo)) fallback_med = true; // (degenerate bucket: generic select)
            else {
                const float hm = fmaxf(c - c_lo, c_hi - c) * 1.0001f;
                hw = __uint_as_float(__float_as_uint(hm) + 2u);
            }
        }
        if (!fallback_med && !(g_ablate & 2)) predicted = bs_predict_mad(bs, wlo, k1, c, w0, P, Q);
        __syncthreads();
        if (tid == 0) { bs->ncollect = 0; bs->nmad = 0; bs->kmin = 0xffffffffu; bs->kmax = 0u; }
        __syncthreads();
    }
    phase(1);
    // ---- pass B: variance + median bucket + MAD bracket --------------------------------------
    sp.key = wlo + (uint32_t)bin; sp.c = c; sp.hw = hw; sp.do_mad = (predicted && !(g_ablate & 16384)) ? 1 : 0;
    if (sp.do_mad) { sp.P = P; sp.Q = Q; } else { sp.P = __builtin_inff(); sp.Q = -1.0f; } // (no bracket: nothing flagged for it)
    if (g_ablate & 4096) { o.sd = 0; o.med = 0; o.mad = 0; return o; }
    SumAux p2 = (have_medmad || fallback_med || (g_ablate & (2048 | 8192))) ? block_np_sum<SIDE_NONE>(x, n, 2, o.mean, bs, sp)
                                              : block_np_sum<SIDE_COLLECT>(x, n, 2, o.mean, bs, sp);
    o.sd = sqrtf(p2.sum / (float)n);
    phase(2);
    if (have_medmad || (g_ablate & (2048 | 8192 | 131072))) { o.med = med_in; o.mad = mad_in; return o; }
    if (fallback_med) {
        __syncthreads();
        if (tid < 64) { float m_ = wave_median(x, n, 0, 0.f, &bs->u.ws); if (tid == 0) bs->bcast[1] = m_; }
        __syncthreads();
        o.med = bs->bcast[1];
        __syncthreads();
    } else if (bs->ncollect > BS_MEDCAP) {
        // more samples in the bucket than the list holds: many equal values (quantised data)
        const uint32_t kmin = bs->kmin, kmax = bs->kmax;
        __syncthreads();
        if (kmin == kmax) { // ... all of them the same value: nothing to select
            const float vk = key2f(kmin);
            o.med = vk;
            if ((n & 1) == 0) {
                float lo = vk;
                if (rk == 0) lo = key2f(bs_max_key_below(x, n, (wlo + (uint32_t)bin) << BS_KSH, bs)); // (rare)
                o.med = (lo + vk) / 2.0f;
            }
        } else o.med = bs_median_dense_bucket(x, n, bs, (wlo + (uint32_t)bin) << BS_KSH, rk);
        if (tid == 0 && n >= 8192) atomicAdd(&g_dbg[kmin == kmax ? 20 : 21], 1ull);
    } else {
        uint32_t below_key = 0;
        if ((n & 1) == 0 && rk == 0) below_key = bs_max_key_below(x, n, (wlo + (uint32_t)bin) << BS_KSH, bs); // (rare)
#ifdef ADP_PHASE_TIMING
        { long long t = clock64(); if (tid == 0) { atomicAdd(&g_dbg[37], (unsigned long long)(t - tph)); if ((n & 1) == 0 && rk == 0) atomicAdd(&g_dbg[38], 1ull); atomicAdd(&g_dbg[39], (unsigned long long)bs->ncollect); } }
#endif
        o.med = bs_median_from_bucket(bs, bs_collect(bs), bs->ncollect, n, rk, below_key);
    }
    phase(3);
    if (g_ablate & 32768) { o.mad = 0.f; return o; }
    // ---- MAD inside the bracket, if it can be proven ---------------------------------------------
    bool done = false;
    if (predicted && !fallback_med) {
        const int M = bs->nmad;
        const int rel = k1 - (int)p2.aux2;
        const bool need_prev = (n & 1) == 0;
        __syncthreads();
        if (M <= BS_MADCAP && rel >= (need_prev ? 1 : 0) && rel < M) {
            float vk, vkm1;
            block_select2_lds((const LDS float *)bs->hist, M, rel, 1, o.med, bs, vk, vkm1); // (M <= BS_MADCAP)
            if (tid < 64) {
                const float lo = need_prev ? vkm1 : vk;
                // |x - med| and |x - c| differ by at most |med - c|: a selected value that far (plus slack) inside
                // [P, Q] cannot be overtaken by a sample counted as "closer" or dropped as "farther"
                // (med itself may sit outside the bucket when n is even: use the actual offset |med - c|)
                const float dm = fabsf(o.med - c) + 0.25f * w0;
                const bool proven = (lo >= P + dm) && (vk <= Q - dm) && (dm < 2.0f * w0);
                if (tid == 0) { bs->bcast[2] = need_prev ? (vkm1 + vk) / 2.0f : vk; bs->flag = proven ? 1 : 0; }
            }
            __syncthreads();
            if (bs->flag) { o.mad = bs->bcast[2]; done = true; }
            __syncthreads();
        }
    }
#ifdef ADP_PHASE_TIMING
    if (tid == 0 && n >= 8192) { atomicMax(&g_dbg[14], (unsigned long long)bs->nmad); atomicMax(&g_dbg[15], (unsigned long long)bs->ncollect);
                                 atomicAdd(&g_dbg[16], (unsigned long long)bs->nmad); atomicAdd(&g_dbg[17], (unsigned long long)bs->ncollect);
                                 if (bs->nmad > 2048) atomicAdd(&g_dbg[18], 1ull); if (bs->nmad > 3072) atomicAdd(&g_dbg[19], 1ull); }
#endif
    if (tid == 0 && n >= 8192 && !(g_ablate & 262144)) { // tallies for the large segments only
        unsigned long long *tl = g_bs_tally[blockIdx.x & (ADP_NTALLY - 1)]; // (a line of its own per 1 / ADP_NTALLY of the workgroups)
        atomicAdd(&tl[0], 1ull);
        if (done) atomicAdd(&tl[1], 1ull);
        if (fallback_med) atomicAdd(&tl[2], 1ull);
        if (!predicted) atomicAdd(&tl[3], 1ull);
        if (predicted && !done && bs->nmad > BS_MADCAP) atomicAdd(&tl[4], 1ull);
    }
    phase(4);
    if (!done) o.mad = bs_mad_two_pass(x, n, bs, o.med, o.sd);
    phase(5);
    return o;
}

// per-read partition request left behind by k_validate
struct PartReq {
    int32_t valid;      // 0: skip (dropped minibatch or exception row)
    int32_t S;          // samples available: min(full_len, m)
    int64_t a_s, a_e, p_e;
    float adapter_med, adapter_mad;
    int32_t have_adapter_medmad;
    int32_t p_none;     // polya_end is None (mvs_detect_overwrite): poly(A) keeps its start only, the RNA partition is all None
};

// the three partitions of read r
template <class ROW, class BS>
static __device__ __forceinline__ void bs_partitions_of_read(const ROW sig, const PartReq &q, adp_row *row, LDS BS *bs)
{
    const int S = q.S;
    unsigned long long present = 0;
    const long long starts[3] = {q.a_s, q.a_e, q.p_e};
    const long long ends[3] = {q.a_e, q.p_e, (long long)S};
    const int c_start[3] = {ADP_C_ADAPTER_START, ADP_C_POLYA_START, ADP_C_RNA_START};
    const int c_len[3] = {ADP_C_ADAPTER_LEN, ADP_C_POLYA_LEN, ADP_C_RNA_LEN};
    for (int p = 0; p < 3; p++) {
        const long long st = starts[p], en = ends[p];
        if (p == 2 && q.p_none) continue;
        if (threadIdx.x == 0) row->col[c_start[p]] = (double)st;
        present |= 1ull << c_start[p];
        if (en <= st || (p == 1 && q.p_none)) continue;
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

// grid = n_reads workgroups of NT threads, WPE waves per SIMD (256 x 5: five workgroups per CU; 512 x 4: two; 1024 x 4: one)
template <class SIG, int NT, int WPE>
__global__ void __launch_bounds__(NT, WPE) k_partition_stats(SIG sigs, int m, const PartReq *__restrict__ req, adp_row *__restrict__ rows)
{
    typedef BlockScratchT<NT> BS;
    extern __shared__ __attribute__((aligned(16))) unsigned char bs_mem_[];
    LDS BS *bs = (LDS BS *)bs_mem_;
    const int r = blockIdx.x;
    const PartReq q = req[r];
    if (!q.valid) return;
    if (threadIdx.x == 0) bs->tail_cached = -1;
    __syncthreads();
    bs_partitions_of_read(sigs.row(r, m), q, rows + r, bs);
}

// (Round 3, measured and dropped: a wave per SHORT read -- numpy-ordered sums and radix selects of wave_stats.h, no workgroup
// barriers -- for reads of at most 16 k samples.  On Pareto lengths that kernel took 42 ms for the reads it relieved this one of
// 3 ms for (profiles/r03_tried_and_dropped.txt): a single wave's scalar loads and four-pass selects are far slower per
// sample than this workgroup's staged passes, however short the segment.)
