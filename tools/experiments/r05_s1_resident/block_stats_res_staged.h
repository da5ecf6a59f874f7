// block_stats_res.h -- S1 calc_partition_stats of a LONG RNA partition with most of the segment kept ON CHIP between its two passes.
//
// k_partition_stats (block_stats.h) reads a segment twice from HBM -- pass A: numpy-ordered sum + bucket histogram, pass B: squared
// deviations + the copies for median and MAD -- with five 256-thread workgroups per CU: a gigabyte is in flight between a segment's
// passes and nothing on the chip holds a byte of it.  Here ONE persistent 512-thread workgroup per CU (256 registers per lane)
// walks the listed reads; of every read each wave KEEPS the first KR slabs (1024 samples = 16 registers per lane) it streams in
// registers and the next KL in LDS, and pass B reads from HBM only what is left, last streamed first (one workgroup per CU keeps
// the footprint in flight under the 256 MB Infinity Cache).  The same lane holds the same samples in both passes and a slab is
// summed by bs_slab_sum either way: numpy's association is untouched, the rows are the two-pass kernel's bit for bit.
//   float32 rows: 10 + 3 of a wave's ~24 slabs at the 200 k window stay on chip (54 %);
//   int16 rows (2 bytes per sample): 20 + 6 -- the whole segment: S1 is ONE pass over HBM.
// The serial phases (bucket search and MAD prediction between the passes; the selections behind pass B) would leave the CU's
// memory pipes idle with a single workgroup on it: pass B's first slabs are requested BEFORE the phase between the passes, and
// the NEXT read's resident slabs are requested before the selections -- into the registers pass B has just emptied -- so that the
// CU streams through them (tools/resident_bw.hip: 7.9 -> 5.0 ms per 32 000 segments for the bare passes, serial phases of 3 + 6 us
// hidden entirely).  Nothing is CALLED while slabs sit in registers (a call would spill them): the helpers are the inlined forms.
// Only the fast path lives here.  Whatever needs another look at the segment (a NaN, a window that missed, a bucket of many
// values, the lower median on a bucket's first sample, a MAD bracket that does not prove itself) is put on a list and left to
// k_partition_stats in its list mode: rare, and slower there than it would be here, never different.
//
// reference: adapted/partition/signal_partitions.py:81-96 (np.mean, np.std, np.median, np.median(|x - med|) of signal[polya_end:]).
#pragma once
#include "block_stats.h"

#define RS_THREADS 512
#define RS_NW (RS_THREADS / 64)
#define RS_LONG_MIN 65536 // shortest RNA partition taken (samples): one numpy chunk per wave

template <class ROW> struct ResShape;
template <> struct ResShape<RowF32> { enum { KR = 7, KL = 3, PF = 2 }; };
template <> struct ResShape<RowI16> { enum { KR = 12, KL = 6, PF = 2 }; };

typedef BlockScratchT<RS_THREADS, true> ResScratch;

template <class RAW4, int KL>
struct ResShared {
    ResScratch bs;
    float slabsum[BS_MAXCHUNK * 8]; // sums of the whole slabs, by slab index (numpy chunk c = slabs 8c .. 8c + 7)
    int item_next, pad_[3];
    RAW4 lres[RS_NW * KL * 4 * 64]; // the LDS-resident slabs, raw, lane-major: [wave][slot][u][lane]
};

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
    typedef typename X::Raw4 Raw4;
    typedef typename X::Raw4L Raw4L;
    typedef typename X::Raw1 Raw1;
    constexpr int KR = ResShape<X>::KR, KL = ResShape<X>::KL, PF = ResShape<X>::PF, NW = RS_NW;
    typedef ResShared<Raw4L, KL> Sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_mem_[];
    LDS Sh *sh = (LDS Sh *)rs_mem_;
    LDS ResScratch *bs = &sh->bs;
    const int tid = threadIdx.x, w = tid >> 6, ln = tid & 63;
    LDS float *wstage = bs->u.stage + w * 8 * BS_LEAF_STRIDE;
    const int count = cnt->n_list;
    int cur = blockIdx.x;
    if (cur >= count) return;
    int nxt = cur + (int)gridDim.x; // the first two items are dealt statically, the later ones by the counter
    if (tid == 0) bs->tail_cached = -1;
    __syncthreads();

    Raw4 keep[KR][4];
    // the RNA partition of a listed read: signal[min(polya_end, S) : S]
    auto describe = [&](int item, X &xo, int &no, int &ro) {
        ro = list[item];
        const long long S = req[ro].S, pe = req[ro].p_e;
        const long long a = pe < S ? pe : S;
        no = (int)(S - a);
        xo = sigs.row(ro, m) + a;
    };
    // the three pivot samples, then this wave's first KR slabs (global slab w + NW q).  Every load is issued whatever the length (a slab
    // that does not exist reads slab w again): straight-line code, so that the waits on these registers are counted, not drained
    auto request = [&](const X &xo, int no, Raw1 (&pv)[3]) {
        pv[0] = xo.raw1(no / 4); pv[1] = xo.raw1(no / 2); pv[2] = xo.raw1((3 * (long long)no) / 4);
        const int nslab = (no / 8192) * 8;
#pragma unroll
        for (int q = 0; q < KR; q++) {
            const int g = w + NW * q;
            const long long p = (long long)(g < nslab ? g : w) * 1024;
#pragma unroll
            for (int u = 0; u < 4; u++) keep[q][u] = xo.raw4u_in(p + (u * 64 + ln) * 4);
        }
    };

    X x, xn;
    int n, r, nn, rn;
    Raw1 pv[3], pvn[3];
    describe(cur, x, n, r);
    request(x, n, pv);
#ifdef ADP_PHASE_TIMING
    long long tph_ = wall_clock64();
#endif
    for (;;) {
        const int nchunk = n / 8192, nslab = nchunk * 8;
        const int myslabs = nslab > w ? (nslab - w + NW - 1) / NW : 0; // whole slabs of this wave: global slabs w, w + NW, ...
        const int nstream = myslabs > KR ? myslabs - KR : 0;           // ... of which those behind the first KR are streamed
        const int tail = n - nchunk * 8192;
        const X xtail = x + (long long)nchunk * 8192;
        const int k1 = n / 2;
        auto slab_pos = [&](int q) { return (long long)(w + NW * q) * 1024; };
        // ---- pass A: numpy-ordered sum + bucket histogram -------------------------------------------------
        uint32_t wlo;
        {
            const float a = x.cook1(pv[0]), b = x.cook1(pv[1]), c3 = x.cook1(pv[2]);
            const float pivot = fmaxf(fminf(a, b), fminf(fmaxf(a, b), c3));
            const uint32_t kb = f2key(pivot) >> BS_KSH;
            wlo = kb >= BS_BINS / 2 ? kb - BS_BINS / 2 : 0u; // centred on the pivot
        }
        for (int i = tid; i < BS_BINS + 4; i += RS_THREADS) bs->hist[i] = 0;
        __syncthreads();
        RS_PHASE(0);
        SideParam sp; sp.key = wlo; sp.c = 0.f; sp.P = 0.f; sp.Q = 0.f; sp.hw = 0.f; sp.do_mad = 0;
        uint32_t aux = 0, aux2 = 0;
        Raw4 pf[PF][4];
#pragma unroll
        for (int d = 0; d < PF; d++)
            if (d < nstream) {
                const long long p = slab_pos(KR + d);
#pragma unroll
                for (int u = 0; u < 4; u++) pf[d][u] = x.raw4uc_in(p + (u * 64 + ln) * 4);
            }
#pragma unroll
        for (int q = 0; q < KR; q++)
            if (q < myslabs) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = x.cook4(keep[q][u]);
                const float s_ = bs_slab_sum<SIDE_HIST>(v, 0, 0.f, bs, wstage, sp, aux, aux2);
                if (ln == 0) sh->slabsum[w + NW * q] = s_;
            }
        RS_PHASE(1);
        for (int i = 0; i < nstream; i++) {
            Raw4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; u++) raw[u] = pf[0][u];
#pragma unroll
            for (int d = 0; d + 1 < PF; d++)
#pragma unroll
                for (int u = 0; u < 4; u++) pf[d][u] = pf[d + 1][u];
            if (i + PF < nstream) {
                const long long p = slab_pos(KR + i + PF);
#pragma unroll
                for (int u = 0; u < 4; u++) pf[PF - 1][u] = x.raw4uc_in(p + (u * 64 + ln) * 4);
            }
            if (i < KL) {
#pragma unroll
                for (int u = 0; u < 4; u++) sh->lres[((w * KL + i) * 4 + u) * 64 + ln] = X::to_lds(raw[u]);
            }
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = x.cook4(raw[u]);
            const float s_ = bs_slab_sum<SIDE_HIST>(v, 0, 0.f, bs, wstage, sp, aux, aux2);
            if (ln == 0) sh->slabsum[w + NW * (KR + i)] = s_;
        }
        RS_PHASE(2);
        float rt = 0.f;
        if (tail > 0) rt = bs_ragged_sum<SIDE_HIST>(xtail, tail, 0, 0.f, bs, sp, aux, aux2);
        RS_PHASE(3);
        // pass B's first streamed slabs (the last ones of pass A) are requested now: they arrive during the phase between the passes
        const int nsb = nstream > KL ? nstream - KL : 0; // slabs pass B streams: KR + KL .. myslabs - 1, walked from the end
#pragma unroll
        for (int d = 0; d < PF; d++)
            if (d < nsb) {
                const long long p = slab_pos(KR + nstream - 1 - d);
#pragma unroll
                for (int u = 0; u < 4; u++) pf[d][u] = x.raw4u_in(p + (u * 64 + ln) * 4);
            }
        __syncthreads();
        // numpy's order above the slabs: a chunk's eight slab sums pairwise, the chunks in sequence, the ragged chunk last
        auto fold = [&](float ragged) {
            if (tid < nchunk) {
                const LDS float *t = sh->slabsum + 8 * tid;
                bs->chunk_sum[tid] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            }
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
        const float mean = fold(rt) / (float)n;
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
        if (tid == 0) { bs->ncollect = 0; bs->nmad = 0; bs->kmin = 0xffffffffu; bs->kmax = 0u; bs->cntb = 0; }
        __syncthreads();
        // ---- pass B: squared deviations + the median's bucket + the MAD bracket -----------------------------
        sp.key = wlo + (uint32_t)bin; sp.c = c; sp.hw = hw; sp.do_mad = 1; sp.P = P; sp.Q = Q;
        aux = 0; aux2 = 0;
        float sd = 0.f;
        RS_PHASE(4);
        if (!bail) {
            rt = 0.f;
            if (tail > 0) rt = bs_ragged_sum<SIDE_COLLECT>(xtail, tail, 2, mean, bs, sp, aux, aux2);
            RS_PHASE(5);
            for (int j = 0; j < nsb; j++) { // streamed slabs, last first
                const int i = nstream - 1 - j;
                Raw4 raw[4];
#pragma unroll
                for (int u = 0; u < 4; u++) raw[u] = pf[0][u];
#pragma unroll
                for (int d = 0; d + 1 < PF; d++)
#pragma unroll
                    for (int u = 0; u < 4; u++) pf[d][u] = pf[d + 1][u];
                if (j + PF < nsb) {
                    const long long p = slab_pos(KR + i - PF);
#pragma unroll
                    for (int u = 0; u < 4; u++) pf[PF - 1][u] = x.raw4u_in(p + (u * 64 + ln) * 4);
                }
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = x.cook4(raw[u]);
                const float s_ = bs_slab_sum<SIDE_COLLECT>(v, 2, mean, bs, wstage, sp, aux, aux2);
                if (ln == 0) sh->slabsum[w + NW * (KR + i)] = s_;
            }
            RS_PHASE(6);
            const int nl = nstream < KL ? nstream : KL;
            for (int i = 0; i < nl; i++) { // the slabs kept in LDS
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const Raw4L t_ = sh->lres[((w * KL + i) * 4 + u) * 64 + ln]; v[u] = x.cook4(X::from_lds(t_)); }
                const float s_ = bs_slab_sum<SIDE_COLLECT>(v, 2, mean, bs, wstage, sp, aux, aux2);
                if (ln == 0) sh->slabsum[w + NW * (KR + i)] = s_;
            }
            RS_PHASE(7);
#pragma unroll
            for (int q = 0; q < KR; q++) // the slabs kept in registers
                if (q < myslabs) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = x.cook4(keep[q][u]);
                    const float s_ = bs_slab_sum<SIDE_COLLECT>(v, 2, mean, bs, wstage, sp, aux, aux2);
                    if (ln == 0) sh->slabsum[w + NW * q] = s_;
                }
        }
        // ---- the registers are free: the next read's resident slabs are requested before the selections ---------------
        RS_PHASE(8);
        unsigned int ticket = 0;
        if (tid == 0) ticket = atomicAdd(&cnt->work, 1u);
        const bool have_next = nxt < count;
        describe(have_next ? nxt : cur, xn, nn, rn);
        request(xn, nn, pvn);
        RS_PHASE(9);
        if (!bail) {
            const uint32_t w2 = (uint32_t)wave_sum((int)aux2);
            if (ln == 0 && w2) __hip_atomic_fetch_add(&bs->cntb, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            sd = sqrtf(fold(rt) / (float)n);
        }
        // ---- median inside its bucket ----------------------------------------------------------------------
        float med = 0.f, mad = 0.f;
        if (!bail) {
            const int ncol = bs->ncollect;
            const uint32_t kmin = bs->kmin, kmax = bs->kmax;
            const bool first_of_even = (n & 1) == 0 && rk == 0; // the lower median is the largest sample BELOW the bucket: another pass
            __syncthreads();
            if (first_of_even) bail = true;
            else if (ncol > BS_MEDCAP) {
                if (kmin == kmax) { // (quantised data: the bucket holds one value)
                    const float vk = key2f(kmin);
                    med = vk;
                    if ((n & 1) == 0) med = (vk + vk) / 2.0f;
                } else bail = true;
            } else med = bs_median_from_bucket_i(bs, bs_collect(bs), ncol, n, rk, 0u);
        }
        // ---- MAD inside the bracket, if that can be proven ---------------------------------------------------
        if (!bail) {
            const int M = bs->nmad;
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
