// validate_wg.h -- V1-V4 (validate_boundaries and what it calls: reference adapted/detect/combined.py:358-631,
// anomalies.py:15-35, real_range.py:33-63, mvs.py:45-158) by a WORKGROUP per read, the slices staged in LDS (round 4).
//
// k_validate (validate.h) gives a read to ONE wave, which walks a chain of ~12 exact selections over global memory, four
// passes each: 348 KB fetched per read for ~60 KB of distinct samples, 0.6-0.7 of the wave's life spent waiting.  Here 256
// threads take the read: every slice under analysis (the adapter, the window around the adapter end, the poly(A) slice, its
// two moving-window series) is copied ONCE into LDS with coalesced loads and its order statistics are taken there --
// vw_select: smallest / largest key, a 1024-bin histogram over that span, a block-wide prefix sum that places all the ranks
// asked for (a median pair, both percentile pairs: one call), the few elements of the ranks' bins copied to short lists
// and the rank inside a bin finished by a bit-wise search.  Sums follow numpy's order (the leaves and tree of wave_stats.h).
//
// This kernel is the FAST path only.  Whatever does not fit its plan -- a slice beyond VW_CAP samples, a NaN inside a slice
// (bottleneck's NaN windows, np.nanmedian), mvs_detect_overwrite, candidates without prepared statistics, windows longer
// than a numpy chunk -- is left untouched and flagged (todo[r] = 1): k_validate runs behind it over the flagged reads.  Same
// operations on the same values: rows are byte-identical to k_validate's.
#pragma once
#include "validate.h"

#define VW_THREADS 256
#define VW_CAP 8192   // samples of a staged slice
#define VW_HB 10
#define VW_NB (1 << VW_HB)
#define VW_MAXQ 6
#define VW_LCAP 128
#define VW_WIN 26    // key window of a selection's histogram: 2^26 keys = 8 octaves below the largest element
// why a read is left to k_validate (todo[r]; adp_debug_fetch what = 9)
#define VW_WHY_SIZE 1   // a slice longer than VW_CAP (or empty), a window beyond a numpy chunk
#define VW_WHY_NAN 2    // a NaN inside a staged slice
#define VW_WHY_LIST 3   // a rank's bin with more distinct elements than a list holds
#define VW_WHY_EXC 4    // an exception row of the reference
#define VW_WHY_SERIES 5 // no prepared moving-window series for a candidate that needs them

struct VwSh {
    float buf[VW_CAP];
    uint32_t hist[VW_NB];
    uint32_t list[VW_MAXQ][VW_LCAP];
    uint32_t wsum[4];
    uint32_t mn, mx;
    int nan, over;
    int q_bin[VW_MAXQ], q_before[VW_MAXQ], q_fill[VW_MAXQ];
    uint32_t q_min[VW_MAXQ], q_max[VW_MAXQ];
    float q_val[VW_MAXQ];
    float leaf[WS_LEAFBUF];
    float fbc[4];
    int ibc[8];
};

// x[0 .. n) -> buf (n <= VW_CAP); true if a NaN is among them (every thread gets the same answer)
template <class ROW>
static __device__ __forceinline__ bool vw_stage(ROW x, int n, LDS VwSh *sh)
{
    const int tid = threadIdx.x;
    if (tid == 0) sh->nan = 0;
    __syncthreads(); // (also: nobody reads the previous contents of buf any more)
    bool bad = false;
    for (int base = 0; base < n; base += VW_THREADS * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + u * VW_THREADS + tid; v[u] = ld_if(x, i, i < n); }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + u * VW_THREADS + tid; if (i < n) { sh->buf[i] = v[u]; bad |= v[u] != v[u]; } }
    }
    if (__any(bad) && (tid & 63) == 0) sh->nan = 1;
    __syncthreads();
    return sh->nan != 0;
}

// The order statistics ks[0 .. NK) (a rank < 0: not asked for) of xform(x[0 .. n)) for an LDS array without NaNs, by all 256
// threads -> sh->q_val[].  False when a rank's bin holds more distinct elements than a list takes (the caller hands the read
// to k_validate); equal elements need no list.
template <int NK>
static __device__ __noinline__ bool vw_select(const LDS float *x, int n, int mode, float c, const int (&ks)[NK], LDS VwSh *sh)
{
    static_assert(NK <= VW_MAXQ, "queries");
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    for (int i = tid; i < VW_NB; i += VW_THREADS) sh->hist[i] = 0;
    if (tid < NK) { sh->q_bin[tid] = -1; sh->q_before[tid] = 0; sh->q_fill[tid] = 0; sh->q_min[tid] = 0xffffffffu; sh->q_max[tid] = 0u; sh->q_val[tid] = 0.f; }
    if (tid == 0) { sh->mn = 0xffffffffu; sh->mx = 0u; sh->over = 0; }
    __syncthreads();
    {
        uint32_t mn = 0xffffffffu, mx = 0u;
        for (int i = tid; i < n; i += VW_THREADS) { const uint32_t k = f2key(ws_xform(x[i], mode, c)); mn = k < mn ? k : mn; mx = k > mx ? k : mx; }
        mn = wave_min(mn); mx = wave_max(mx);
        if (ln == 0) {
            __hip_atomic_fetch_min(&sh->mn, mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(&sh->mx, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    // the bins cover at most the top 2^VW_WIN keys below the largest one (8 octaves): |x - med| reaches down to 0, and the octaves
    // of float keys near 0 would take all the bins; whatever lies below goes to bin 0, which then settles no rank
    const uint32_t mn0 = sh->mn;
    const uint32_t mn = (sh->mx - mn0 > ((1u << VW_WIN) - 1u)) ? sh->mx - ((1u << VW_WIN) - 1u) : mn0;
    const bool clamped = mn != mn0;
    const uint32_t span = sh->mx - mn;
    const int bits = span ? 32 - __clz(span) : 0;
    const int s = bits > VW_HB ? bits - VW_HB : 0; // bin = (key - mn) >> s < 2^VW_HB
    for (int i = tid; i < n; i += VW_THREADS) {
        const uint32_t k = f2key(ws_xform(x[i], mode, c));
        const uint32_t d = (k > mn ? k : mn) - mn;
        __hip_atomic_fetch_add(&sh->hist[d >> s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    {   // block-wide prefix sum: thread t owns bins 4t .. 4t + 3
        const uint32_t h0 = sh->hist[4 * tid], h1 = sh->hist[4 * tid + 1], h2 = sh->hist[4 * tid + 2], h3 = sh->hist[4 * tid + 3];
        const uint32_t sum = h0 + h1 + h2 + h3;
        const uint32_t incl = (uint32_t)wave_scan_incl((int)sum);
        if (ln == 63) sh->wsum[wv] = incl;
        __syncthreads();
        uint32_t excl = incl - sum;
        for (int w = 0; w < VW_THREADS / 64; w++) { const uint32_t t = sh->wsum[w]; excl += w < wv ? t : 0u; }
#pragma unroll
        for (int q = 0; q < NK; q++) {
            const int k = ks[q];
            if (k >= 0 && (uint32_t)k >= excl && (uint32_t)k < excl + sum) {
                uint32_t acc = excl; int bin;
                if ((uint32_t)k < acc + h0) bin = 4 * tid;
                else { acc += h0; if ((uint32_t)k < acc + h1) bin = 4 * tid + 1;
                    else { acc += h1; if ((uint32_t)k < acc + h2) bin = 4 * tid + 2; else { acc += h2; bin = 4 * tid + 3; } } }
                sh->q_bin[q] = bin; sh->q_before[q] = (int)acc;
            }
        }
    }
    __syncthreads();
    int qb[NK], slot[NK];
#pragma unroll
    for (int q = 0; q < NK; q++) { qb[q] = ks[q] >= 0 ? sh->q_bin[q] : -2; slot[q] = q; }
#pragma unroll
    for (int q = NK - 1; q > 0; q--)
#pragma unroll
        for (int p = 0; p < q; p++) if (qb[p] == qb[q]) slot[q] = slot[q] < p ? slot[q] : p; // (the first query of a bin keeps its list)
    const uint32_t lowmask = s ? ((1u << s) - 1u) : 0u;
    for (int i = tid; i < n; i += VW_THREADS) {
        const uint32_t k = f2key(ws_xform(x[i], mode, c));
        const uint32_t d = (k > mn ? k : mn) - mn;
        const int b = (int)(d >> s);
#pragma unroll
        for (int q = 0; q < NK; q++) {
            if (b == qb[q] && slot[q] == q) {
                const uint32_t low = d & lowmask;
                __hip_atomic_fetch_min(&sh->q_min[q], low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&sh->q_max[q], low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int pos = __hip_atomic_fetch_add(&sh->q_fill[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (pos < VW_LCAP) sh->list[q][pos] = low;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NK; q++) {
        if ((q & 3) != wv) continue; // a wave per query
        const int k = ks[q];
        if (k < 0) continue;
        const int sl = slot[q];
        uint32_t low;
        const uint32_t lmin = sh->q_min[sl], lmax = sh->q_max[sl];
        const int total = sh->q_fill[sl];
        if (qb[q] < 0 || (clamped && qb[q] == 0)) { if (ln == 0) sh->over = 1; continue; } // (no bin: cannot happen on a NaN-free array; bin 0 of a clamped range is open-ended)
        if (lmin == lmax) low = lmin;
        else if (total > VW_LCAP) { if (ln == 0) sh->over = 1; continue; }
        else {
            const int krem = k - sh->q_before[q];
            uint32_t pre = 0;
            for (int bit = s - 1; bit >= 0; bit--) {
                const uint32_t trial = pre | (1u << bit);
                int cnt = 0;
                for (int i0 = 0; i0 < total; i0 += 64) {
                    const int i = i0 + ln;
                    const uint32_t e = i < total ? sh->list[sl][i] : 0xffffffffu;
                    cnt += __popcll(__ballot(e < trial));
                }
                if (cnt <= krem) pre = trial;
            }
            low = pre;
        }
        if (ln == 0) sh->q_val[q] = key2f(mn + ((uint32_t)qb[q] << s) + low);
    }
    __syncthreads();
    return sh->over == 0;
}

// np.median of an LDS array (NaN-free).  ok = false: the read goes to k_validate.
static __device__ __forceinline__ float vw_median(const LDS float *x, int n, int mode, float c, LDS VwSh *sh, bool &ok)
{
    if (n <= 0) return __builtin_nanf("");
    const int ks[2] = {n / 2, (n & 1) ? -1 : n / 2 - 1};
    ok &= vw_select<2>(x, n, mode, c, ks, sh);
    const float vk = sh->q_val[0], vkm1 = sh->q_val[1];
    return (n & 1) ? vk : (vkm1 + vk) / 2.0f;
}

// np.percentile(x, 85) - np.percentile(x, 15) (linear; the arithmetic of wave_percentile_t), both from one selection
static __device__ __forceinline__ double vw_local_range(const LDS float *x, int n, LDS VwSh *sh, bool &ok)
{
    int lo85, hi85, lo15, hi15; double g85, g15;
    cs_pct_ranks(n, 85.0, lo85, hi85, g85);
    cs_pct_ranks(n, 15.0, lo15, hi15, g15);
    const int ks[4] = {hi85, hi85 == lo85 ? -1 : hi85 - 1, hi15, hi15 == lo15 ? -1 : hi15 - 1};
    ok &= vw_select<4>(x, n, 0, 0.f, ks, sh);
    return cs_pct_value(sh->q_val[0], sh->q_val[1], lo85, hi85, g85) - cs_pct_value(sh->q_val[2], sh->q_val[3], lo15, hi15, g15);
}

// np.add.reduce(xform(x[0 .. n))) in float32 with numpy's association, n < 8192 (one ragged chunk: the leaves and the tree of
// ws_enum_leaves / ws_eval_tree), by wave 0; every thread of the workgroup gets the sum.
static __device__ __noinline__ float vw_np_sum(const LDS float *x, int n, int mode, float c, LDS VwSh *sh)
{
    if (threadIdx.x < 64) {
        const int ln = threadIdx.x;
        int id = 0;
        {   // leaves, left to right (numpy: a node longer than 128 splits into (n2, len - n2), n2 = len / 2 rounded down to 8)
            int st_off[16], st_len[16], sp = 1;
            st_off[0] = 0; st_len[0] = n;
            while (sp > 0) {
                sp--;
                int off = st_off[sp], len = st_len[sp];
                while (len > 128) {
                    int n2 = len / 2;
                    n2 -= n2 % 8;
                    st_off[sp] = off + n2; st_len[sp] = len - n2; sp++;
                    len = n2;
                }
                if ((id & 63) == ln) {
                    const LDS float *p = x + off;
                    sh->leaf[id] = pw_leaf_f32(len, [&](int i) { return ws_xform(p[i], mode, c); });
                }
                id++;
            }
        }
        ws_sync();
        float ret = 0.0f;
        {   // the same tree over the leaf sums, post-order
            int st_len[16], st_state[16], sp = 1, id2 = 0;
            float st_left[16];
            st_len[0] = n; st_state[0] = 0;
            while (sp > 0) {
                const int t = sp - 1, len = st_len[t];
                if (len <= 128) { ret = sh->leaf[id2++]; sp--; continue; }
                int n2 = len / 2;
                n2 -= n2 % 8;
                if (st_state[t] == 0) { st_state[t] = 1; st_len[sp] = n2; st_state[sp] = 0; sp++; }
                else if (st_state[t] == 1) { st_left[t] = ret; st_state[t] = 2; st_len[sp] = len - n2; st_state[sp] = 0; sp++; }
                else { ret = st_left[t] + ret; sp--; }
            }
        }
        if (ln == 0) sh->fbc[0] = ret;
    }
    __syncthreads();
    const float out = sh->fbc[0];
    __syncthreads();
    return out;
}

// grid: persistent workgroups (blockIdx.x strides over the reads); block = VW_THREADS; dynamic LDS = sizeof(VwSh).
// todo[r] = 1: the read was NOT handled (k_validate takes it); 0: its row and partition request are complete.
template <class SIG>
__global__ void __launch_bounds__(VW_THREADS) k_validate_wg(ValidateInT<SIG> in, adp_cfg cfg, adp_row *__restrict__ rows, PartReq *__restrict__ preq,
                                                            int8_t *__restrict__ todo)
{
    extern __shared__ unsigned char vw_raw[];
    LDS VwSh *sh = (LDS VwSh *)vw_raw;
    const int tid = threadIdx.x, ln = tid & 63;
    for (int r = blockIdx.x; r < in.n_reads; r += gridDim.x) {
        adp_row *row = rows + r;
        __syncthreads();
        {
            uint32_t *w = reinterpret_cast<uint32_t *>(row);
            for (int i = tid; i < (int)(sizeof(adp_row) / 4); i += VW_THREADS) w[i] = 0;
        }
        __syncthreads();
        if (tid == 0) { row->n_cand = -1; row->n_open_pores = -1; row->open_pores_more = -1; preq[r].valid = 0; todo[r] = 0; }
        if (in.mbs && in.mbs[r / in.mbsize].status != ADP_MB_OK) continue; // dropped minibatch: zero row
        const typename SIG::Row sig = in.sig.row(r, in.m);
        const long long full_len = in.full_len[r];
        const int S = (int)(full_len < in.m ? full_len : in.m);
        const int64_t *bd = in.bounds + (size_t)r * (1 + in.kmax);
        const long long a_in = bd[0];
        const long long p_in = in.kmax > 0 ? bd[1] : 0;
        const bool topk_none = in.topk_none ? in.topk_none[r] != 0 : false;
        long long a_s = 0;
        const long long a_e = a_in;
        long long p_best = p_in;
        int success = 1, fail = ADP_F_NONE, mvs_mask = 0;
        float adapter_med = 0.f, adapter_mad = 0.f;
        bool have_med = false, ok = true, exception = false;
        int defer = 0; // != 0: why the read is left to k_validate (VW_WHY_*; adp_debug_fetch what = 9)
        unsigned long long present = 0ull;
        int n_open = -1;
        auto set = [&](int c, double v) { if (tid == 0) row->col[c] = v; present |= 1ull << c; };

        // ---- the adapter: median, MAD, open pores, real-range check -- all on ONE staged copy of signal[0 : adapter_end)
        if (a_e == 0) { success = 0; fail = ADP_F_NO_ADAPTER; }
        else {
            const int b = (int)(a_e < S ? a_e : S);
            if (b > VW_CAP || b <= 0) defer = VW_WHY_SIZE;
            else if (vw_stage(sig, b, sh)) defer = VW_WHY_NAN;
            else {
                adapter_med = vw_median(sh->buf, b, 0, 0.f, sh, ok);
                adapter_mad = vw_median(sh->buf, b, 1, adapter_med, sh, ok);
                have_med = true;
            }
        }
        if (!defer && !ok) defer = VW_WHY_LIST;
        if (!defer && success && have_med && adapter_mad != 0.0f &&
            !in_range_d((double)adapter_mad, cfg.adapter_mad_range[0], cfg.adapter_mad_range[1])) {
            success = 0; fail = ADP_F_ADAPTER_MAD;
        }
        if (!defer && success && cfg.detect_open_pores) {
            // V2 by wave 0 over the staged adapter: positions >= 200 pA; keep pos[i] (i >= 1) with a gap >= 10 to pos[i-1]; none kept -> [pos[-1]]
            const int b = (int)(a_e < S ? a_e : S);
            if (tid < 64) {
                int npos = 0, nvalid = 0, prev_last = -1, lastpos = -1, lastvalid = -1;
                for (int base = 0; base < b; base += 64) {
                    const int i = base + ln;
                    const bool f = (i < b) && (200.0f <= sh->buf[i]);
                    const unsigned long long mk = __ballot(f);
                    if (mk) {
                        const unsigned long long lower = mk & ((1ull << ln) - 1ull);
                        const int prev = lower ? (base + 63 - __clzll((long long)lower)) : prev_last;
                        const bool first_overall = (npos == 0) && (lower == 0);
                        const bool valid = f && !first_overall && (i - prev >= 10);
                        const unsigned long long vm = __ballot(valid);
                        if (valid) {
                            const int slot = nvalid + __popcll(vm & ((1ull << ln) - 1ull));
                            if (slot < ADP_MAX_OPEN_PORES) row->open_pores[slot] = i;
                        }
                        if (vm) lastvalid = base + 63 - __clzll((long long)vm);
                        nvalid += __popcll(vm);
                        npos += __popcll(mk);
                        lastpos = base + 63 - __clzll((long long)mk);
                        prev_last = lastpos;
                    }
                }
                int no = 0; long long last = -1;
                if (npos == 0) no = 0;
                else if (npos == 1 || nvalid == 0) { no = 1; last = lastpos; if (ln == 0) row->open_pores[0] = lastpos; }
                else { no = nvalid; last = lastvalid; }
                if (no > ADP_MAX_OPEN_PORES) {
                    // the reference's list has no length limit: the whole of it goes to the call's arena (second scan, rare)
                    unsigned int off = 0;
                    if (ln == 0) off = atomicAdd(in.op_used, (unsigned int)no);
                    off = __shfl(off, 0);
                    if ((unsigned long long)off + (unsigned long long)no <= in.op_cap) {
                        int np2 = 0, nv2 = 0, pl2 = -1;
                        for (int base = 0; base < b; base += 64) {
                            const int i = base + ln;
                            const bool f = (i < b) && (200.0f <= sh->buf[i]);
                            const unsigned long long mk = __ballot(f);
                            if (mk) {
                                const unsigned long long lower = mk & ((1ull << ln) - 1ull);
                                const int prev = lower ? (base + 63 - __clzll((long long)lower)) : pl2;
                                const bool first_overall = (np2 == 0) && (lower == 0);
                                const bool valid = f && !first_overall && (i - prev >= 10);
                                const unsigned long long vm = __ballot(valid);
                                if (valid) in.op_arena[off + nv2 + __popcll(vm & ((1ull << ln) - 1ull))] = i;
                                nv2 += __popcll(vm);
                                np2 += __popcll(mk);
                                pl2 = base + 63 - __clzll((long long)mk);
                            }
                        }
                        if (ln == 0) row->open_pores_more = (int32_t)off;
                    } else if (ln == 0) row->open_pores_more = -2; // (the host grows the arena and runs the kernels again)
                }
                if (ln == 0) { sh->ibc[0] = no; sh->ibc[1] = (int)last; }
            }
            __syncthreads();
            n_open = sh->ibc[0];
            if (n_open > 0) {
                a_s = sh->ibc[1];
                if (a_e - a_s < cfg.min_obs_adapter) { success = 0; fail = ADP_F_OPEN_PORE; }
            }
            __syncthreads();
        }
        if (!defer && success && cfg.real_signal_check) {
            const int a = (int)(a_s < S ? a_s : S), b = (int)(a_e < S ? a_e : S);
            int n = b - a; if (n < 0) n = 0;
            const LDS float *x = (const LDS float *)sh->buf + a;
            bool rok = false;
            if (n >= 2 * cfg.mean_window) {
                if (cfg.mean_window >= 8192 || cfg.mean_window < 1) defer = VW_WHY_SIZE;
                else {
                    const float ms = vw_np_sum(x, cfg.mean_window, 0, 0.f, sh) / (float)cfg.mean_window;
                    const float me = vw_np_sum(x + n - cfg.mean_window, cfg.mean_window, 0, 0.f, sh) / (float)cfg.mean_window;
                    set(ADP_C_REAL_MEAN_START, (double)ms);
                    set(ADP_C_REAL_MEAN_END, (double)me);
                    if (in_range_d((double)ms, cfg.mean_start_range[0], cfg.mean_start_range[1]) &&
                        in_range_d((double)me, cfg.mean_end_range[0], cfg.mean_end_range[1])) {
                        const int k = n < cfg.max_obs_local_range ? n : cfg.max_obs_local_range;
                        const double lr = vw_local_range(x + n - k, k, sh, ok);
                        set(ADP_C_REAL_LOCAL_RANGE, lr);
                        rok = in_range_d(lr, cfg.local_range[0], cfg.local_range[1]);
                    }
                }
            }
            if (!rok) { success = 0; fail = ADP_F_REAL_RANGE; }
        }
        if (!defer && !ok) defer = VW_WHY_LIST;
        // ---- the poly(A) candidates (V4)
        if (!defer && success && cfg.mvs_detect_check) {
            if (p_best == 0) { success = 0; fail = ADP_F_NO_POLYA; }
            else {
                double pr0 = cfg.pA_mean_range[0], pr1 = cfg.pA_mean_range[1];
                int exc = 0;
                if (range_empty(cfg.pA_mean_range) && !range_empty(cfg.pA_mean_adapter_med_scale_range)) {
                    pr0 = cfg.pA_mean_adapter_med_scale_range[0] * (double)adapter_med;
                    pr1 = cfg.pA_mean_adapter_med_scale_range[1] * (double)adapter_med;
                } else if (range_empty(cfg.pA_mean_range)) exc = ADP_F_EXC_PA_RANGE;
                if (!exc && topk_none) exc = ADP_F_EXC_TOPK_NONE;
                if (exc) defer = VW_WHY_EXC; // (exception rows are k_validate's business: rare)
                long long p_series = 0; // the poly(A) end the precomputed series reach
                if (in.series && in.have_series[r])
                    for (int c = 0; c < in.kmax; c++) { const long long pc = bd[1 + c]; if (pc == 0) break; if (pc > p_series) p_series = pc; }
                float shift_val = 0.f; bool shift_have = false;
                for (int c = 0; !defer && c < in.kmax; c++) {
                    const long long p_e = bd[1 + c];
                    if (p_e == 0) break;
                    // mvs_check (validate.h) with the statistics taken in LDS
                    int mok = 0, vec_fail = 31;
                    double o_mean = 0.0, o_var = 0.0, o_med = 0.0, o_lr = 0.0, o_shift = 0.0;
                    const bool early = (p_e == 0 || a_e == 0 || p_e < a_e || p_e - a_e <= 2) || ((long long)S < a_e + cfg.median_shift_window);
                    if (!early) {
                        const int a = (int)(a_e < S ? a_e : S), b = (int)(p_e < S ? p_e : S);
                        const int n = b - a;
                        const bool wvar = !(p_e - a_e <= cfg.pA_var_window + 2), wmean = !(p_e - a_e <= cfg.pA_mean_window + 2);
                        if ((wvar && (cfg.pA_var_window > n || cfg.pA_var_window < 1)) || (wmean && (cfg.pA_mean_window > n || cfg.pA_mean_window < 1))) { defer = VW_WHY_EXC; break; }
                        float fvar, fmean, fmed;
                        double lrange;
                        const CandStat *cst = in.cstat ? in.cstat + (size_t)r * in.kmax + c : nullptr;
                        if (cst && cst->ready && wvar && wmean) { fvar = cst->fvar; fmean = cst->fmean; fmed = cst->fmed; lrange = cst->q85 - cst->q15; }
                        else {
                            const bool pre = p_series > 0 && p_e <= p_series;
                            if (n <= 0 || n > VW_CAP || ((wvar || wmean) && !pre)) { defer = VW_WHY_SERIES; break; }
                            const float *pm = in.series + (size_t)r * 2 * in.series_cap, *pv = pm + in.series_cap;
                            if (wvar) { if (vw_stage(as_row(pv), n - cfg.pA_var_window + 1, sh)) { defer = VW_WHY_NAN; break; } fvar = vw_median(sh->buf, n - cfg.pA_var_window + 1, 0, 0.f, sh, ok); }
                            if (wmean) { if (vw_stage(as_row(pm), n - cfg.pA_mean_window + 1, sh)) { defer = VW_WHY_NAN; break; } fmean = vw_median(sh->buf, n - cfg.pA_mean_window + 1, 0, 0.f, sh, ok); }
                            if (vw_stage(sig + a, n, sh)) { defer = VW_WHY_NAN; break; }
                            if (!wvar) { const float mu = vw_np_sum(sh->buf, n, 0, 0.f, sh) / (float)n; fvar = vw_np_sum(sh->buf, n, 2, mu, sh) / (float)n; }
                            if (!wmean) fmean = vw_np_sum(sh->buf, n, 0, 0.f, sh) / (float)n;
                            {   // median and both percentiles of the slice from ONE selection
                                int lo85, hi85, lo15, hi15; double g85, g15;
                                cs_pct_ranks(n, 85.0, lo85, hi85, g85);
                                cs_pct_ranks(n, 15.0, lo15, hi15, g15);
                                const int ks[6] = {n / 2, (n & 1) ? -1 : n / 2 - 1, hi85, hi85 == lo85 ? -1 : hi85 - 1, hi15, hi15 == lo15 ? -1 : hi15 - 1};
                                ok &= vw_select<6>(sh->buf, n, 0, 0.f, ks, sh);
                                fmed = (n & 1) ? sh->q_val[0] : (sh->q_val[1] + sh->q_val[0]) / 2.0f;
                                lrange = cs_pct_value(sh->q_val[2], sh->q_val[3], lo85, hi85, g85) - cs_pct_value(sh->q_val[4], sh->q_val[5], lo15, hi15, g15);
                            }
                        }
                        if (!shift_have) {
                            long long r1 = a_e + cfg.median_shift_window; if (r1 > S) r1 = S;
                            long long l0 = a_e - cfg.median_shift_window; if (l0 < 0) l0 = 0;
                            const int nl = (int)(a - l0), nr = (int)(r1 - a);
                            if (nl + nr > VW_CAP || nl <= 0 || nr <= 0) { defer = VW_WHY_SIZE; break; }
                            if (vw_stage(sig + l0, nl + nr, sh)) { defer = VW_WHY_NAN; break; }
                            const float right = vw_median(sh->buf + nl, nr, 0, 0.f, sh, ok);
                            const float left = vw_median(sh->buf, nl, 0, 0.f, sh, ok);
                            shift_val = right - left;
                            shift_have = true;
                        }
                        if (!ok) { defer = VW_WHY_LIST; break; }
                        o_mean = (double)fmean; o_var = (double)fvar; o_med = (double)fmed; o_lr = lrange; o_shift = (double)shift_val;
                        int f = 0;
                        if (!in_range_d(o_mean, pr0, pr1)) f |= 1;
                        if (!in_range_d(o_var, cfg.pA_var_range[0], cfg.pA_var_range[1])) f |= 2;
                        if (!in_range_d(o_med, cfg.polyA_med_range[0], cfg.polyA_med_range[1])) f |= 4;
                        if (!in_range_d(o_lr, cfg.polyA_local_range[0], cfg.polyA_local_range[1])) f |= 8;
                        if (!in_range_d(o_shift, cfg.median_shift_range[0], cfg.median_shift_range[1])) f |= 16;
                        vec_fail = f; mok = (f == 0);
                    }
                    set(ADP_C_MVS_MEAN, o_mean); set(ADP_C_MVS_VAR, o_var);
                    set(ADP_C_MVS_POLYA_MED, o_med); set(ADP_C_MVS_LOCAL_RANGE, o_lr);
                    set(ADP_C_MVS_MED_SHIFT, o_shift);
                    if (!mok) {
                        success = 0; // never reset: later candidates only refresh the reported values
                        if (o_mean == 0) { fail = ADP_F_MVS_NOT_ENOUGH; mvs_mask = 0; }
                        else { fail = ADP_F_MVS_CHECKS; mvs_mask = vec_fail; }
                    }
                    if (success) { p_best = p_e; break; }
                }
            }
        }
        if (!defer && success && cfg.detect_med_shift) {
            const long long w = cfg.med_shift_window;
            long long r1 = a_e + w; if (r1 > full_len) r1 = full_len; if (r1 > S) r1 = S;
            const long long a = a_e < S ? a_e : S;
            long long l0 = a_e - w; if (l0 < 0) l0 = 0; if (l0 > S) l0 = S;
            const int nl = (int)(a - l0), nr = (int)(r1 - a);
            if (nl + nr > VW_CAP || nl <= 0 || nr <= 0) defer = VW_WHY_SIZE;
            else if (vw_stage(sig + l0, nl + nr, sh)) defer = VW_WHY_NAN;
            else {
                const float right = vw_median(sh->buf + nl, nr, 0, 0.f, sh, ok);
                const float left = vw_median(sh->buf, nl, 0, 0.f, sh, ok);
                const float shv = right - left;
                if (!ok) defer = VW_WHY_LIST;
                else {
                    set(ADP_C_MED_SHIFT, (double)shv);
                    if (!in_range_d((double)shv, cfg.med_shift_range[0], cfg.med_shift_range[1])) { success = 0; fail = ADP_F_MED_SHIFT; }
                }
            }
        }
        if (defer || exception) { if (tid == 0) todo[r] = (int8_t)(defer ? defer : VW_WHY_EXC); continue; }
        // S1 partition statistics are computed by k_partition_stats (one 256-thread block per read)
        if (tid == 0) {
            PartReq q;
            q.valid = 1; q.S = S; q.a_s = a_s; q.a_e = a_e; q.p_e = p_best;
            q.adapter_med = adapter_med; q.adapter_mad = adapter_mad;
            q.have_adapter_medmad = (have_med && a_s == 0 && a_e == a_in) ? 1 : 0; q.p_none = 0;
            preq[r] = q;
        }
        set(ADP_C_ADAPTER_END, (double)a_e);
        set(ADP_C_POLYA_END, (double)p_best);
        set(ADP_C_SIGNAL_LEN, (double)full_len);
        set(ADP_C_PRELOADED, (double)S);
        set(ADP_C_PRIMARY_ADAPTER_END, (double)a_in);
        set(ADP_C_PRIMARY_POLYA_END, (double)p_in);
        if (tid == 0) {
            if (!topk_none) {
                const int nc = in.kmax < ADP_MAX_CAND ? in.kmax : ADP_MAX_CAND;
                row->n_cand = nc;
                for (int c = 0; c < nc; c++) row->cand[c] = bd[1 + c];
            }
            row->n_open_pores = n_open;
            row->present = present;
            row->success = success;
            row->fail_code = fail;
            row->mvs_fail_mask = mvs_mask;
        }
    }
}
