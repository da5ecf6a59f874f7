// wave_stats.h -- exact order statistics and numpy-ordered float32 reductions over a segment
// of one read, cooperatively by the 64 lanes of one wave (blockDim.x == 64).
//
// These restate what the reference obtains from numpy on float32 slices (np.median,
// np.percentile(linear), np.mean/np.var/np.std: reference adapted/partition/
// signal_partitions.py:81-96, adapted/detect/mvs.py:88-129, adapted/detect/real_range.py:46-58)
// so that the device reproduces the same float32 values:
//   * selection: MSB-first 8-bit radix select on order-preserving keys, 4 passes over the
//     segment (L2/MALL resident between passes), 256-bin histogram in LDS with a wave-uniform
//     fast path (one add when all 64 lanes hit the same bin, the common case for the top byte).
//     The (k-1)-th value comes out of the same 4 passes (needed for even-count medians and for
//     percentile interpolation).
//   * sums: numpy's add.reduce order (8192-element chunks accumulated in sequence; each chunk a
//     balanced pairwise tree over 128-element leaves summed with 8 interleaved accumulators).
//     A full chunk is staged through LDS in two halves; lane l sums leaf l and the tree is a
//     xor-butterfly.  Ragged tail chunks follow the same recursion generically.
#pragma once
#include "common.h"

#define WS_LEAF_STRIDE 136
#define WS_STAGE_FLOATS (8 * WS_LEAF_STRIDE)
#define WS_LEAFBUF 160

// barrier among the lanes of ONE wave that orders its LDS traffic (LDS operations of one wave execute in order)
static __device__ __forceinline__ void ws_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct WaveScratch {
    union { // (the selections' histogram and the sums' staging rows are never in use together: 4992 bytes per wave, 32 waves per CU)
        uint32_t hist[256];
        float stage[WS_STAGE_FLOATS];
    };
    float leaf[WS_LEAFBUF];
};

// Optional LDS copy of the segment under analysis (k_validate): the many order statistics taken from
// one adapter / poly(A) slice then cost ONE pass over global memory instead of four each.
#define WS_SEG_CAP 6144
struct SegCache {
    float seg[WS_SEG_CAP];
    uintptr_t src;    // address of the global range currently mirrored: n samples from there
    int n;
};

// transform applied on load: mode 0: x; mode 1: |x - c| (float32); mode 2: (x - c)^2 (float32)
static __device__ __forceinline__ float ws_xform(float x, int mode, float c)
{
    if (mode == 0) return x;
    float d = x - c;
    return mode == 1 ? fabsf(d) : d * d;
}

// ---------------------------------------------------------------- selection
// x_(k) and x_(k-1) of xform(x[0..n)), 0 <= k < n.  All lanes return the same values.
// One pass for the smallest and largest key, then an MSB-first radix select over d = key - min, 8 bits of the
// SPAN per pass (a segment spanning 1.5 octaves of pA needs 3 passes, a constant one none): the digits are spread
// over the bins whatever the level of the signal, so plain LDS atomics do.
// SKIPNAN: NaN entries are left out altogether (np.nanmedian's view of the array): k counts among the others.
template <class P, bool SKIPNAN = false>
static __device__ __forceinline__ void wave_select2_impl(P x, int n, int k, int mode, float c, LDS WaveScratch *ws,
                                                         float &vk, float &vkm1)
{
    const int ln = lane_id();
    constexpr int UN = SKIPNAN ? 2 : 8; // loads in flight per lane (the NaN-skipping variant is a rare path: few registers)
    uint32_t mn = 0xffffffffu, mx = 0u;
    bool has_nan = false;
    for (int base = 0; base < n; base += 64 * UN) {
        float v[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) { int i = base + u * 64 + ln; v[u] = ld_or_first(x, i, i < n); }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const float xf = ws_xform(v[u], mode, c);
            if (SKIPNAN && xf != xf) continue;
            has_nan |= xf != xf;
            uint32_t key = f2key(xf); mn = key < mn ? key : mn; mx = key > mx ? key : mx;
        }
    }
    // np.partition sorts NaN behind everything and np.median / np.percentile return NaN when one is present
    // (numpy/lib/_function_base_impl.py _median, _quantile): so do the order statistics here
    if (__any(has_nan)) { vk = __builtin_nanf(""); vkm1 = vk; return; }
    mn = wave_min(mn); mx = wave_max(mx);
    const uint32_t span = mx - mn;
    int rb = span ? 32 - __clz(span) : 0; // bits of d still unresolved
    uint32_t prefix = 0, krem = (uint32_t)k, below = 0; // below: largest d + 1 under the selected prefix (last pass)
    int lowbin = -1, lastw = 0;
    while (rb > 0) {
        const int w = rb < 8 ? rb : 8;
        const int shift = rb - w;
        for (int i = ln; i < 256; i += 64) ws->hist[i] = 0;
        ws_sync();
        for (int base = 0; base < n; base += 64 * UN) {
            float v[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) { int i = base + u * 64 + ln; v[u] = ld_if(x, i, i < n); } // (entries beyond n are not used)
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const int i = base + u * 64 + ln;
                const float xf = ws_xform(v[u], mode, c);
                const uint32_t d = f2key(xf) - mn;
                const uint32_t top = (rb >= 32) ? 0u : (d >> rb);
                if (i < n && !(SKIPNAN && xf != xf)) {
                    if (top == prefix) __hip_atomic_fetch_add(&ws->hist[(d >> shift) & ((1u << w) - 1u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else if (shift == 0 && top < prefix && d + 1u > below) below = d + 1u;
                }
            }
        }
        ws_sync();
        // locate the bin of rank krem: each lane owns 4 consecutive bins
        const uint32_t h0 = ws->hist[4 * ln], h1 = ws->hist[4 * ln + 1], h2 = ws->hist[4 * ln + 2], h3 = ws->hist[4 * ln + 3];
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
        ws_sync();
    }
    vk = key2f(mn + prefix);
    vkm1 = vk;
    if (krem == 0 && k > 0) { // first of its key: the value before it is the largest key below
        below = wave_max(below);
        uint32_t d0 = below ? below - 1u : 0u;
        bool have = below != 0;
        if (span && lowbin >= 0) {
            uint32_t da = (prefix & ~((1u << lastw) - 1u)) | (uint32_t)lowbin;
            if (!have || da > d0) d0 = da;
            have = true;
        }
        if (have) vkm1 = key2f(mn + d0);
    }
}
// The same radix select for NR ranks of ONE array at once (round 4): the key-range pass is shared, and every later pass histograms
// the elements under each rank's prefix into that rank's own 256 bins (NR <= 4 histograms fit the scratch's staging rows) -- the two
// percentiles of a slice, or its median and both percentiles, cost the passes of one selection instead of two or three.
// k[] in any order; vk[r] = x_(k[r]), vkm1[r] = x_(k[r] - 1) as wave_select2_impl returns them (same keys, same rules).
template <class P, int NR>
static __device__ __noinline__ void wave_select_ranks(P x, int n, const int (&k)[NR], int mode, float c, LDS WaveScratch *ws,
                                                      float (&vk)[NR], float (&vkm1)[NR])
{
    static_assert(NR >= 2 && NR * 256 <= WS_STAGE_FLOATS, "the histograms live in the staging rows");
    const int ln = lane_id();
    constexpr int UN = 8;
    LDS uint32_t *hist = (LDS uint32_t *)ws->stage;
    uint32_t mn = 0xffffffffu, mx = 0u;
    bool has_nan = false;
    for (int base = 0; base < n; base += 64 * UN) {
        float v[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) { int i = base + u * 64 + ln; v[u] = ld_or_first(x, i, i < n); }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const float xf = ws_xform(v[u], mode, c);
            has_nan |= xf != xf;
            uint32_t key = f2key(xf); mn = key < mn ? key : mn; mx = key > mx ? key : mx;
        }
    }
    if (__any(has_nan)) {
#pragma unroll
        for (int r = 0; r < NR; r++) { vk[r] = __builtin_nanf(""); vkm1[r] = vk[r]; }
        return;
    }
    mn = wave_min(mn); mx = wave_max(mx);
    const uint32_t span = mx - mn;
    int rb = span ? 32 - __clz(span) : 0;
    uint32_t prefix[NR], krem[NR], below[NR];
    int lowbin[NR], lastw = 0;
#pragma unroll
    for (int r = 0; r < NR; r++) { prefix[r] = 0; krem[r] = (uint32_t)k[r]; below[r] = 0; lowbin[r] = -1; }
    while (rb > 0) {
        const int w = rb < 8 ? rb : 8;
        const int shift = rb - w;
        for (int i = ln; i < 256 * NR; i += 64) hist[i] = 0;
        ws_sync();
        for (int base = 0; base < n; base += 64 * UN) {
            float v[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) { int i = base + u * 64 + ln; v[u] = ld_if(x, i, i < n); }
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const int i = base + u * 64 + ln;
                const float xf = ws_xform(v[u], mode, c);
                const uint32_t d = f2key(xf) - mn;
                const uint32_t top = (rb >= 32) ? 0u : (d >> rb);
                const uint32_t dig = (d >> shift) & ((1u << w) - 1u);
                if (i < n) {
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        if (top == prefix[r]) __hip_atomic_fetch_add(&hist[256 * r + dig], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else if (shift == 0 && top < prefix[r] && d + 1u > below[r]) below[r] = d + 1u;
                    }
                }
            }
        }
        ws_sync();
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const LDS uint32_t *h = hist + 256 * r + 4 * ln;
            const uint32_t h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3];
            const int s = (int)(h0 + h1 + h2 + h3);
            const int incl = wave_scan_incl(s);
            const int excl = incl - s;
            const bool mine = (int)krem[r] >= excl && (int)krem[r] < incl;
            int bin = 0, before = 0;
            if (mine) {
                int cacc = excl;
                if ((int)krem[r] < cacc + (int)h0) { bin = 4 * ln; before = cacc; }
                else { cacc += h0;
                    if ((int)krem[r] < cacc + (int)h1) { bin = 4 * ln + 1; before = cacc; }
                    else { cacc += h1;
                        if ((int)krem[r] < cacc + (int)h2) { bin = 4 * ln + 2; before = cacc; }
                        else { cacc += h2; bin = 4 * ln + 3; before = cacc; } } }
            }
            const unsigned long long mm = __ballot(mine);
            const int src = __ffsll((long long)mm) - 1;
            bin = __shfl(bin, src);
            before = __shfl(before, src);
            if (shift == 0) {
                int cand = -1;
                if (4 * ln < bin && h0) cand = 4 * ln;
                if (4 * ln + 1 < bin && h1) cand = 4 * ln + 1;
                if (4 * ln + 2 < bin && h2) cand = 4 * ln + 2;
                if (4 * ln + 3 < bin && h3) cand = 4 * ln + 3;
                lowbin[r] = wave_max(cand);
            }
            prefix[r] = (prefix[r] << w) | (uint32_t)bin;
            krem[r] -= (uint32_t)before;
        }
        if (shift == 0) lastw = w;
        rb = shift;
        ws_sync();
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
        vk[r] = key2f(mn + prefix[r]);
        vkm1[r] = vk[r];
        if (krem[r] == 0 && k[r] > 0) {
            const uint32_t bl = wave_max(below[r]);
            uint32_t d0 = bl ? bl - 1u : 0u;
            bool have = bl != 0;
            if (span && lowbin[r] >= 0) {
                uint32_t da = (prefix[r] & ~((1u << lastw) - 1u)) | (uint32_t)lowbin[r];
                if (!have || da > d0) d0 = da;
                have = true;
            }
            if (have) vkm1[r] = key2f(mn + d0);
        }
    }
}

// x: a signal row (RowF32 / RowI16, common.h) or a plain float array in global memory
template <class X>
static __device__ __noinline__ void wave_select2_global(X x, int n, int k, int mode, float c,
                                                        LDS WaveScratch *ws, float &vk, float &vkm1)
{
    wave_select2_impl<X>(x, n, k, mode, c, ws, vk, vkm1);
}
// x_(k), x_(k-1) of the non-NaN entries of x[0..n); 0 <= k < their count
static __device__ __noinline__ void wave_select2_skipnan(const float *x, int n, int k, int mode, float c,
                                                         LDS WaveScratch *ws, float &vk, float &vkm1)
{
    wave_select2_impl<RowF32, true>(as_row(x), n, k, mode, c, ws, vk, vkm1);
}
static __device__ __noinline__ void wave_select2_lds(const LDS float *x, int n, int k, int mode, float c, LDS WaveScratch *ws,
                                                     float &vk, float &vkm1)
{
    wave_select2_impl<const LDS float *>(x, n, k, mode, c, ws, vk, vkm1);
}

// x_(k), x_(k-1) of xform(x[0..n)); with a SegCache the slice is mirrored into LDS first (reused while the
// requested range stays inside the mirrored one)
template <class X>
static __device__ void wave_select2(X x, int n, int k, int mode, float c, LDS WaveScratch *ws, float &vk,
                                    float &vkm1, LDS SegCache *sc = nullptr)
{
    if (sc && n <= WS_SEG_CAP) {
        const uintptr_t src = sc->src;
        const int cn = sc->n;
        long long d = 0;
        bool hit = false;
        if (src && x.key() >= src) { d = (long long)(x.key() - src) / (long long)sizeof(*x.p); hit = d + n <= cn; }
        if (!hit) {
            ws_sync();
            for (int i = lane_id(); i < n; i += 64) sc->seg[i] = x[i];
            if (lane_id() == 0) { sc->src = x.key(); sc->n = n; }
            ws_sync();
            d = 0;
        }
        wave_select2_lds(sc->seg + d, n, k, mode, c, ws, vk, vkm1);
    } else {
        wave_select2_global<X>(x, n, k, mode, c, ws, vk, vkm1);
    }
}

// np.median(xform(x[0..n))) for a NaN-free float32 segment
template <class X>
static __device__ __noinline__ float wave_median_t(X x, int n, int mode, float c, LDS WaveScratch *ws, LDS SegCache *sc = nullptr)
{
    if (n <= 0) return __builtin_nanf("");
    float vk, vkm1;
    wave_select2(x, n, n / 2, mode, c, ws, vk, vkm1, sc);
    if (n & 1) return vk;
    return (vkm1 + vk) / 2.0f;
}

// one np.percentile(x, q) value (linear method): virtual index (n-1)*q/100 in float64,
// diff in float32, interpolation in float64 (numpy/lib/_function_base_impl.py _lerp)
template <class X>
static __device__ __forceinline__ float wave_median(X x, int n, int mode, float c, LDS WaveScratch *ws, LDS SegCache *sc = nullptr)
{
    const auto r = as_row(x); // (a plain float pointer becomes a RowF32)
    return wave_median_t<decltype(as_row(x))>(r, n, mode, c, ws, sc);
}

template <class X>
static __device__ __noinline__ double wave_percentile_t(X x, int n, double q100, LDS WaveScratch *ws, LDS SegCache *sc = nullptr)
{
    double q = q100 / 100.0;
    double vi = (double)(n - 1) * q;
    int lo = (int)floor(vi);
    if (lo < 0) lo = 0;
    if (lo > n - 1) lo = n - 1;
    int hi = min(lo + 1, n - 1);
    double g = vi - (double)lo;
    float vk, vkm1;
    wave_select2(x, n, hi, 0, 0.0f, ws, vk, vkm1, sc);
    float a = (hi == lo) ? vk : vkm1;
    float b = vk;
    float diff = b - a;
    double r = (double)a + (double)diff * g;
    if (g >= 0.5) r = (double)b - (double)diff * (1.0 - g);
    return r;
}

template <class X>
static __device__ __forceinline__ double wave_percentile(X x, int n, double q100, LDS WaveScratch *ws, LDS SegCache *sc = nullptr)
{
    const auto r = as_row(x);
    return wave_percentile_t<decltype(as_row(x))>(r, n, q100, ws, sc);
}

// np.percentile's rank arithmetic (wave_percentile_t's), apart from the selection
static __device__ __forceinline__ void ws_pct_ranks(int n, double q100, int &lo, int &hi, double &g)
{
    const double vi = (double)(n - 1) * (q100 / 100.0);
    lo = (int)floor(vi);
    if (lo < 0) lo = 0;
    if (lo > n - 1) lo = n - 1;
    hi = min(lo + 1, n - 1);
    g = vi - (double)lo;
}
static __device__ __forceinline__ double ws_pct_value(float vk, float vkm1, int lo, int hi, double g)
{
    const float a = (hi == lo) ? vk : vkm1, b = vk;
    const float diff = b - a;
    double r = (double)a + (double)diff * g;
    if (g >= 0.5) r = (double)b - (double)diff * (1.0 - g);
    return r;
}
// np.percentile(x, 85) - np.percentile(x, 15) of a NaN-free slice (n >= 1): one multi-rank selection
template <class X>
static __device__ double wave_local_range(X x0, int n, LDS WaveScratch *ws)
{
    const auto x = as_row(x0);
    int lo85, hi85, lo15, hi15; double g85, g15;
    ws_pct_ranks(n, 85.0, lo85, hi85, g85);
    ws_pct_ranks(n, 15.0, lo15, hi15, g15);
    const int k[2] = {hi85, hi15};
    float vk[2], vkm1[2];
    wave_select_ranks<decltype(as_row(x0)), 2>(x, n, k, 0, 0.0f, ws, vk, vkm1);
    return ws_pct_value(vk[0], vkm1[0], lo85, hi85, g85) - ws_pct_value(vk[1], vkm1[1], lo15, hi15, g15);
}
// np.median(x) and the same local range: three ranks of one slice
template <class X>
static __device__ void wave_median_local_range(X x0, int n, LDS WaveScratch *ws, float &med, double &lrange)
{
    const auto x = as_row(x0);
    int lo85, hi85, lo15, hi15; double g85, g15;
    ws_pct_ranks(n, 85.0, lo85, hi85, g85);
    ws_pct_ranks(n, 15.0, lo15, hi15, g15);
    const int k[3] = {n / 2, hi85, hi15};
    float vk[3], vkm1[3];
    wave_select_ranks<decltype(as_row(x0)), 3>(x, n, k, 0, 0.0f, ws, vk, vkm1);
    med = (n & 1) ? vk[0] : (vkm1[0] + vk[0]) / 2.0f;
    lrange = ws_pct_value(vk[1], vkm1[1], lo85, hi85, g85) - ws_pct_value(vk[2], vkm1[2], lo15, hi15, g15);
}

// ---------------------------------------------------------------- numpy-ordered sums
// numpy's pairwise recursion, iteratively: a node longer than 128 splits into (n2, len - n2) with
// n2 = (len/2) rounded down to a multiple of 8.  Leaves are visited left to right.
template <class X>
static __device__ __noinline__ void ws_enum_leaves(X x, int off0, int len0, int mode, float c, LDS WaveScratch *ws, int &id)
{
    int st_off[16], st_len[16];
    int sp = 0;
    st_off[0] = off0; st_len[0] = len0; sp = 1;
    while (sp > 0) {
        sp--;
        int off = st_off[sp], len = st_len[sp];
        while (len > 128) {
            int n2 = len / 2;
            n2 -= n2 % 8;
            st_off[sp] = off + n2; st_len[sp] = len - n2; sp++; // right child later
            len = n2;                                           // descend left
        }
        if ((id & 63) == lane_id()) {
            const X p = x + off;
            ws->leaf[id] = pw_leaf_f32(len, [&](int i) { return ws_xform(p[i], mode, c); });
        }
        id++;
    }
}

// post-order evaluation of the same tree over the leaf sums (every lane computes the same value)
static __device__ __noinline__ float ws_eval_tree(int len0, const LDS WaveScratch *ws, int &id)
{
    // explicit stack of (len, state, left value): state 0 = visit left, 1 = visit right, 2 = combine
    int st_len[16]; int st_state[16]; float st_left[16];
    int sp = 0;
    st_len[0] = len0; st_state[0] = 0; sp = 1;
    float ret = 0.0f;
    while (sp > 0) {
        int t = sp - 1;
        int len = st_len[t];
        if (len <= 128) { ret = ws->leaf[id++]; sp--; continue; }
        int n2 = len / 2;
        n2 -= n2 % 8;
        if (st_state[t] == 0) { st_state[t] = 1; st_len[sp] = n2; st_state[sp] = 0; sp++; }
        else if (st_state[t] == 1) { st_left[t] = ret; st_state[t] = 2; st_len[sp] = len - n2; st_state[sp] = 0; sp++; }
        else { ret = st_left[t] + ret; sp--; }
    }
    return ret;
}

// np.add.reduce(xform(x[0..n))) in float32 with numpy's association
template <class X>
static __device__ __noinline__ float wave_np_sum_t(X x, int n, int mode, float c, LDS WaveScratch *ws)
{
    const int ln = lane_id();
    float total = 0.0f;
    for (int s = 0; s < n; s += 8192) {
        const int len = min(8192, n - s);
        float chunk;
        if (len == 8192) {
            // eight phases of 8 leaves (1024 samples): lane = (leaf, accumulator) runs one of numpy's 8 interleaved
            // accumulator chains (16 samples); xor-shuffles fold the accumulators, then the 8 leaves (three levels of
            // the balanced tree); the 8 phase sums make the top three levels
            float ph[8];
            const X p = x + s;
#pragma unroll
            for (int phs = 0; phs < 8; phs++) {
                ws_sync();
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = (u * 64 + ln) * 4;
                    const X q4 = p + (phs * 1024 + e);
                    float a0 = ws_xform(q4[0], mode, c), a1 = ws_xform(q4[1], mode, c), a2 = ws_xform(q4[2], mode, c), a3 = ws_xform(q4[3], mode, c);
                    LDS float *d = ws->stage + (e >> 7) * WS_LEAF_STRIDE + (e & 127);
                    d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3;
                }
                ws_sync();
                const LDS float *q = ws->stage + (ln >> 3) * WS_LEAF_STRIDE + (ln & 7);
                float r = q[0];
#pragma unroll
                for (int t = 1; t < 16; t++) r += q[8 * t];
                r = r + __shfl_xor(r, 1);
                r = r + __shfl_xor(r, 2);
                r = r + __shfl_xor(r, 4);
                r = r + __shfl_xor(r, 8);
                r = r + __shfl_xor(r, 16);
                r = r + __shfl_xor(r, 32);
                ph[phs] = r;
            }
            chunk = ((ph[0] + ph[1]) + (ph[2] + ph[3])) + ((ph[4] + ph[5]) + (ph[6] + ph[7]));
        } else {
            int id = 0;
            ws_sync();
            ws_enum_leaves(x + s, 0, len, mode, c, ws, id);
            ws_sync();
            int id2 = 0;
            chunk = ws_eval_tree(len, ws, id2); // every lane evaluates the same small tree
        }
        total += chunk;
    }
    return total;
}

template <class X>
static __device__ __forceinline__ float wave_np_sum(X x, int n, int mode, float c, LDS WaveScratch *ws)
{
    const auto r = as_row(x);
    return wave_np_sum_t<decltype(as_row(x))>(r, n, mode, c, ws);
}

template <class X>
static __device__ float wave_np_mean(X x, int n, LDS WaveScratch *ws) { return wave_np_sum(x, n, 0, 0.0f, ws) / (float)n; }

// np.var: mean in float32, squared deviations in float32, sum / n
template <class X>
static __device__ float wave_np_var(X x, int n, LDS WaveScratch *ws, float *mean_out)
{
    float mu = wave_np_mean(x, n, ws);
    if (mean_out) *mean_out = mu;
    return wave_np_sum(x, n, 2, mu, ws) / (float)n;
}
