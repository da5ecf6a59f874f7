// common.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#define ADP_NDBG 80 // debug tallies (adp_debug_fetch what = 8)
#define ADP_NTALLY 256
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adapted_hip.h"

// pointers into LDS keep their address space across (non-inlined) function boundaries, so that
// accesses stay ds_* instructions instead of degrading to flat_* ones
#define LDS __attribute__((address_space(3)))
// likewise for global memory: a plain pointer handed to a non-inlined function is "generic" and its loads become
// flat_* instructions, which also count against the LDS wait counter (lgkmcnt) and so serialise with LDS traffic
#define GLB __attribute__((address_space(1)))
#define WAVE 64
#define CK 16          // cumulative-sum checkpoint spacing (pooled samples)
#define TRACE_TILE 1024 // pooled samples handled per wave iteration in the gains kernel (64 lanes x CK)
#define SUMBLK 64      // trace summary block (max/min per 64 trace points)

// per-minibatch state of the batch-global normalisation (N1)
struct MbState {
    unsigned long long n_valid;   // non-NaN samples in batch[:, :T]
    unsigned long long krem;      // rank of the upper median inside the current key window
    unsigned long long c_below;   // samples whose key lies below the window (pass 1)
    uint32_t kbase;               // first key of the current window (2^21 keys in pass 1, 2^10 in pass 2)
    int32_t bad;                  // the sampled window missed the rank: aligned 3-pass path takes over
    uint32_t cklo, ckw;           // narrow bracket [cklo, cklo + ckw) whose keys pass 1 copies out (ckw = 0: none)
    int32_t done;                 // the statistic was finished from the copied bracket: pass 2 is skipped
    int32_t fused;                // both statistics were settled by the single fused pass (n1_fused.h)
    float med, mad, lo, hi;       // N1 parameters (float32, as numpy holds them)
    int32_t status;               // ADP_MB_*
    int32_t pad;
};

// streaming loads of samples that are not read again soon: the non-temporal hint (nt) keeps them from displacing
// the caches' useful lines; measured +6 % on a pure streaming read (6.8 vs 6.4 TB/s)
typedef float adp_v4f __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ float4 ld_stream4(const float4 *p)
{
    const adp_v4f v = __builtin_nontemporal_load(reinterpret_cast<const adp_v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// a / d in float32 for MANY numerators and ONE divisor: with y = RN(1/d) (one IEEE division), q0 = RN(a y) is within
// an ulp of the quotient, r = a - d q0 is exact in one fma, and RN(q0 + r y) is the correctly rounded quotient
// (Markstein's theorem; it needs y correctly rounded, which fails only for a significand of all ones, and no
// underflow of r).  fdiv_ok(d) says whether the divisor qualifies; tests/test_gpu_ops.py checks 4e9 quotients.
static __device__ __forceinline__ bool fdiv_ok(float d)
{
    const uint32_t u = __float_as_uint(d), e = (u >> 23) & 0xffu;
    return (u & 0x7fffffu) != 0x7fffffu && e > 40u && e < 215u; // normal, far from the ends of the exponent range
}
static __device__ __forceinline__ float fdiv_shared(float a, float d, float y)
{
    const float q0 = a * y;
    const float r = __builtin_fmaf(-d, q0, a);
    return __builtin_fmaf(r, y, q0);
}

// ---------------------------------------------------------------- the resident signal matrix
// The reference's minibatch layout is float32 pA [n, m], NaN from each read's end on (adapted/file_proc.py:143-190): SigF32.
// SigI16 is the same matrix as the sequencer stores it -- raw int16 ADC samples [n, m] plus a per-read calibration,
// pA = scale * (float32(adc) + offset) (pod5's calibrate_signal_array; both operations rounded to float32, never fused) --
// turned into the float32 value in registers, bit-identical to adp_calibrate_i16's output, so that every streaming pass
// moves 2 bytes per sample instead of 4.  Samples at or beyond min(full_len, m) read as NaN, like the padding they replace.
// Kernels take the matrix type as a template parameter and go through a Row: row[i], row + k (a row that starts k samples
// later), row.f4(q) / row.f4s(q) (samples 4q .. 4q+3 of a row whose start is 16-byte / 8-byte aligned; f4s = streaming
// hint), row.f4u(i) (four samples from any position i).
typedef float adp_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef short adp_s4 __attribute__((ext_vector_type(4)));
typedef short adp_s4u __attribute__((ext_vector_type(4), aligned(2)));

struct RowF32 {
    const GLB float *p;
    __device__ __forceinline__ float operator[](long long i) const { return p[i]; }
    // x[i] where ok, else dflt (a masked load: the compiler keeps several of them in flight)
    __device__ __forceinline__ float at_or(long long i, bool ok, float dflt) const { return ok ? p[i] : dflt; }
    __device__ __forceinline__ RowF32 operator+(long long k) const { return RowF32{p + k}; }
    __device__ __forceinline__ RowF32 operator-(long long k) const { return RowF32{p - k}; }
    __device__ __forceinline__ float4 f4(long long q) const { const adp_v4f v = reinterpret_cast<const GLB adp_v4f *>(p)[q]; return make_float4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ float4 f4s(long long q) const
    {
        const adp_v4f v = __builtin_nontemporal_load(reinterpret_cast<const GLB adp_v4f *>(p) + q);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    __device__ __forceinline__ float4 f4u(long long i) const
    {
        const adp_f4u v = __builtin_nontemporal_load(reinterpret_cast<const GLB adp_f4u *>(p + i));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    // *_in: the caller guarantees that the four samples exist (no padding among them): nothing to check for either row type
    __device__ __forceinline__ float4 f4_in(long long q) const { return f4(q); }
    __device__ __forceinline__ float4 f4s_in(long long q) const { return f4s(q); }
    __device__ __forceinline__ float4 f4u_in(long long i) const { return f4u(i); }
    __device__ __forceinline__ float4 f4uc(long long i) const // (the same through the caches: data that is read again soon)
    {
        const adp_f4u v = *reinterpret_cast<const GLB adp_f4u *>(p + i);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    // raw4u_in / cook4: the load and the conversion of f4u_in apart, for software pipelines that keep the RAW samples in flight
    // (PREFETCH = how many such steps a streaming loop keeps ahead: the same bytes in flight for either row type)
    typedef float4 Raw4;
    static constexpr int PREFETCH = 1;
    __device__ __forceinline__ Raw4 raw4u_in(long long i) const { return f4u(i); }
    __device__ __forceinline__ float4 cook4(const Raw4 &v) const { return v; }
    __device__ __forceinline__ bool vec_ok() const { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
    __device__ __forceinline__ uintptr_t key() const { return reinterpret_cast<uintptr_t>(p); }
    __device__ __forceinline__ long long diff(const RowF32 &o) const { return p - o.p; }
};

struct RowI16 {
    const GLB int16_t *p;
    float sc, of;
    int n; // samples that exist from p on (may be <= 0): the rest is the NaN padding
    __device__ __forceinline__ float cal(short a) const { const float t = (float)a + of; return sc * t; }
    // (the load is unconditional -- a position behind the read's end is still inside the matrix, or at most 6 bytes behind its
    // last row for the vector forms, which the allocations cover -- so that the compiler keeps independent loads in flight
    // instead of branching around each)
    __device__ __forceinline__ float operator[](long long i) const { const float v = cal(p[i]); return i < n ? v : __builtin_nanf(""); }
    // (here ONE unconditional load -- element 0 stands in when !ok -- and a select: a conditional load followed by the
    // conversion makes the compiler branch around each load and wait for it, eight round trips instead of one)
    __device__ __forceinline__ float at_or(long long i, bool ok, float dflt) const { const float v = (*this)[ok ? i : 0]; return ok ? v : dflt; }
    __device__ __forceinline__ RowI16 operator+(long long k) const { return RowI16{p + k, sc, of, (int)(n - k)}; }
    __device__ __forceinline__ RowI16 operator-(long long k) const { return RowI16{p - k, sc, of, (int)(n + k)}; }
    __device__ __forceinline__ float4 conv4(adp_s4 v, long long i) const
    {
        const float nanv = __builtin_nanf("");
        float4 r = make_float4(cal(v.x), cal(v.y), cal(v.z), cal(v.w));
        if (i + 3 >= n) { if (i >= n) r.x = nanv; if (i + 1 >= n) r.y = nanv; if (i + 2 >= n) r.z = nanv; r.w = nanv; }
        return r;
    }
    __device__ __forceinline__ float4 f4(long long q) const { return conv4(reinterpret_cast<const GLB adp_s4 *>(p)[q], 4 * q); }
    __device__ __forceinline__ float4 f4s(long long q) const { return conv4(__builtin_nontemporal_load(reinterpret_cast<const GLB adp_s4 *>(p) + q), 4 * q); }
    __device__ __forceinline__ float4 f4u(long long i) const
    {
        const adp_s4u v = __builtin_nontemporal_load(reinterpret_cast<const GLB adp_s4u *>(p + i));
        const adp_s4 w = {v.x, v.y, v.z, v.w};
        return conv4(w, i);
    }
    __device__ __forceinline__ float4 f4uc(long long i) const
    {
        const adp_s4u v = *reinterpret_cast<const GLB adp_s4u *>(p + i);
        const adp_s4 w = {v.x, v.y, v.z, v.w};
        return conv4(w, i);
    }
    __device__ __forceinline__ float4 cal4(adp_s4 v) const { return make_float4(cal(v.x), cal(v.y), cal(v.z), cal(v.w)); }
    __device__ __forceinline__ float4 f4_in(long long q) const { return cal4(reinterpret_cast<const GLB adp_s4 *>(p)[q]); }
    __device__ __forceinline__ float4 f4s_in(long long q) const { return cal4(__builtin_nontemporal_load(reinterpret_cast<const GLB adp_s4 *>(p) + q)); }
    __device__ __forceinline__ float4 f4u_in(long long i) const
    {
        const adp_s4u v = __builtin_nontemporal_load(reinterpret_cast<const GLB adp_s4u *>(p + i));
        const adp_s4 w = {v.x, v.y, v.z, v.w};
        return cal4(w);
    }
    typedef adp_s4 Raw4;
    static constexpr int PREFETCH = 2; // (3 with 4 workgroups per CU for the registers: slower, 29.5 vs 26.4 ms -- the kernel wants its 5 waves per SIMD)
    __device__ __forceinline__ Raw4 raw4u_in(long long i) const
    {
        const adp_s4u v = __builtin_nontemporal_load(reinterpret_cast<const GLB adp_s4u *>(p + i));
        const adp_s4 w = {v.x, v.y, v.z, v.w};
        return w;
    }
    __device__ __forceinline__ float4 cook4(const Raw4 &v) const { return cal4(v); }
    __device__ __forceinline__ bool vec_ok() const { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
    __device__ __forceinline__ uintptr_t key() const { return reinterpret_cast<uintptr_t>(p); }
    __device__ __forceinline__ long long diff(const RowI16 &o) const { return p - o.p; }
};

struct SigF32 {
    typedef RowF32 Row;
    const float *base;
    __device__ __forceinline__ Row row(long long r, int m) const { return Row{(const GLB float *)base + (size_t)r * m}; }
    __device__ __forceinline__ bool vec_ok(int m) const { return (m & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0; }
};
struct SigI16 {
    typedef RowI16 Row;
    const int16_t *base;
    const float *scale, *offset;
    const int32_t *full_len;
    __device__ __forceinline__ Row row(long long r, int m) const
    {
        const int fl = full_len[r];
        return Row{(const GLB int16_t *)base + (size_t)r * m, scale[r], offset[r], fl < m ? (fl > 0 ? fl : 0) : m};
    }
    __device__ __forceinline__ bool vec_ok(int m) const { return (m & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 7) == 0; }
};
// plain float arrays (series, pooled values, copies) go through the same helpers as rows
// x[i] if ok, else some valid element's value or 0 (the caller does not use it): the load pattern each row type is fastest with
static __device__ __forceinline__ float ld_if(const RowF32 &x, long long i, bool ok) { return ok ? x.p[i] : 0.f; }
static __device__ __forceinline__ float ld_if(const RowI16 &x, long long i, bool ok) { return x[ok ? i : 0]; }
static __device__ __forceinline__ float ld_if(const LDS float *x, long long i, bool ok) { return ok ? x[i] : 0.f; }
// x[i] if ok, else x[0]
static __device__ __forceinline__ float ld_or_first(const RowF32 &x, long long i, bool ok) { return ok ? x.p[i] : x.p[0]; }
static __device__ __forceinline__ float ld_or_first(const RowI16 &x, long long i, bool ok) { return x[ok ? i : 0]; }
static __device__ __forceinline__ float ld_or_first(const LDS float *x, long long i, bool ok) { return ok ? x[i] : x[0]; }
static __device__ __forceinline__ RowF32 as_row(const float *p) { return RowF32{(const GLB float *)p}; }
static __device__ __forceinline__ RowF32 as_row(float *p) { return RowF32{(const GLB float *)p}; }
static __device__ __forceinline__ RowF32 as_row(RowF32 r) { return r; }
static __device__ __forceinline__ RowI16 as_row(RowI16 r) { return r; }

static __device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// order-preserving float32 <-> uint32 key
static __device__ __forceinline__ uint32_t f2key(float f)
{
    // (~u for a negative float, u | 0x80000000 for the others -- as shift / or / xor: the compare-and-select form costs the same
    // three instructions plus the wait states between a v_cmp and the v_cndmask that reads its mask)
    const uint32_t u = __float_as_uint(f);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
static __device__ __forceinline__ float key2f(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

template <class T>
static __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <class T>
static __device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}
template <class T>
static __device__ __forceinline__ T wave_min(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o); v = w < v ? w : v; }
    return v;
}
// inclusive prefix sum across the wave
static __device__ __forceinline__ int wave_scan_incl(int v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int w = __shfl_up(v, o); if (lane_id() >= o) v += w; }
    return v;
}

// numpy's pairwise float32 sum of n <= 128 values read through `get(i)` (sequential in one lane).
// numpy/_core/src/umath/loops_utils.h.src (pairwise_sum): n < 8 plain loop; else 8 accumulators.
template <class F>
static __device__ __forceinline__ float pw_leaf_f32(int n, F get)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res += get(i);
        return res;
    }
    float r0 = get(0), r1 = get(1), r2 = get(2), r3 = get(3), r4 = get(4), r5 = get(5), r6 = get(6), r7 = get(7);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += get(i); r1 += get(i + 1); r2 += get(i + 2); r3 += get(i + 3);
        r4 += get(i + 4); r5 += get(i + 5); r6 += get(i + 6); r7 += get(i + 7);
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += get(i);
    return res;
}
