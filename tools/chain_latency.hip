// What does a dependent float32 instruction cost on gfx950, and at what clock does a nearly idle chip run a handful of waves?
//   hipcc --offload-arch=gfx950 -O2 tools/chain_latency.hip -o /tmp/cl && /tmp/cl
// Per launch shape (workgroups x waves): cycles (s_memtime) and nanoseconds (s_memrealtime, 100 MHz) per step of
//   A  s += t                      (one dependent v_add_f32)
//   B  s += t; s = s < 0 ? 0 : s   (add, compare, select: the clamped sum of squares of bn_move_var as written)
//   C  s = max(s + t, 0)           (add, max)
//   D  eight independent adds      (issue cost)
//   E  four independent packed float32 operations (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32)
//   F  a dependent v_add_f64; G  the two float64 chains of a cumulative sum (a += x, b += x * x); H  G behind its float32 -> float64 conversion
// and the core clock = cycles / time.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(float *sink, long long *stamps, int steps)
{
    float s = threadIdx.x * 1e-3f, t = 1e-3f + blockIdx.x * 1e-9f;
    float u[8] = {s, s + 1, s + 2, s + 3, s + 4, s + 5, s + 6, s + 7};
    typedef float f2 __attribute__((ext_vector_type(2)));
    double d0 = s, d1 = t, d2 = 0.0, d3 = 0.0;
    f2 w[5] = {{s, t}, {t, s}, {s, s}, {t, t}, {1.0f, 1.0001f}};
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < steps; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(t));
            if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1\n\tv_cmp_gt_f32 vcc, 0, %0\n\tv_cndmask_b32 %0, %0, 0, vcc" : "+v"(s) : "v"(t) : "vcc");
            if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1\n\tv_max_f32 %0, %0, 0" : "+v"(s) : "v"(t));
            if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_fma_f32 %3, %3, %4, %4"
                                        : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "v"(w[4]));
            if (MODE == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(d1));
            if (MODE == 6) asm volatile("v_add_f64 %0, %0, %2\n\tv_mul_f64 %3, %2, %2\n\tv_add_f64 %1, %1, %3" : "+v"(d0), "+v"(d2), "+v"(d1), "=v"(d3));
            if (MODE == 7) asm volatile("v_cvt_f64_f32 %2, %4\n\tv_add_f64 %0, %0, %2\n\tv_mul_f64 %3, %2, %2\n\tv_add_f64 %1, %1, %3" : "+v"(d0), "+v"(d2), "=v"(d1), "=v"(d3) : "v"(t));
            if (MODE == 3) asm volatile("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\tv_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                                        : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(t));
        }
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
    float acc = s;
    for (int j = 0; j < 8; j++) acc += u[j];
    for (int j = 0; j < 4; j++) acc += w[j].x + w[j].y;
    acc += (float)(d0 + d1 + d2 + d3);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE> static void run(const char *name, int wgs, int waves, float *sink, long long *stamps)
{
    const int steps = 20000;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(64 * waves), 0, 0, sink, stamps, steps); hipDeviceSynchronize(); }
    long long h[2];
    hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost);
    const double n = (double)steps * 16 * (MODE == 3 ? 8 : MODE == 4 ? 4 : 1);
    printf("%4d wg x %d waves  %-28s %6.2f cycles  %6.2f ns per %s   clock %.2f GHz\n", wgs, waves, name, h[0] / n, h[1] * 10.0 / n, MODE >= 3 ? "instruction" : "step", h[0] / (h[1] * 10.0));
}
int main()
{
    float *sink; long long *stamps;
    hipMalloc(&sink, 4096 * 1024 * 4); hipMalloc(&stamps, 64);
    for (int wgs : {1, 256})
        for (int waves : {1, 6}) {
            run<0>("A add", wgs, waves, sink, stamps);
            run<1>("B add, compare, select", wgs, waves, sink, stamps);
            run<2>("C add, max", wgs, waves, sink, stamps);
            run<3>("D independent adds", wgs, waves, sink, stamps);
            run<4>("E independent packed ops", wgs, waves, sink, stamps);
            run<5>("F f64 add (dependent)", wgs, waves, sink, stamps);
            run<6>("G f64 a += x; b += x * x", wgs, waves, sink, stamps);
            run<7>("H cvt + G (k_cumsum's step)", wgs, waves, sink, stamps);
        }
    return 0;
}
