// f64_rate.hip -- issue rates of the float64 instructions the LLR gains are made of (v_fma_f64, v_add_f64,
// v_mul_f64, v_rcp_f64, v_cvt_f64_i32) on this device, in lane-operations per second, at several occupancies.
// DESIGN.md section 5 prices k_gains against these numbers.
//   hipcc --offload-arch=gfx950 -O3 -o f64_rate tools/f64_rate.hip && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND, int CHAINS>
__global__ void __launch_bounds__(256) k_rate(double *out, int iters, double seed)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = seed + 1e-3 * (threadIdx.x + c);
    const double m = 1.0 + 1e-9 * seed, a = 1e-12 * seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) x[c] = __builtin_fma(x[c], m, a);
                else if (KIND == 1) x[c] = x[c] + a;
                else if (KIND == 2) x[c] = x[c] * m;
                else if (KIND == 3) x[c] = __builtin_amdgcn_rcp(x[c]);
                else if (KIND == 4) x[c] = (double)(int)((long long)__double_as_longlong(x[c]) >> 40);
                else if (KIND == 5) { float f = (float)x[c]; f = __builtin_fmaf(f, 1.0001f, 1e-7f); x[c] = (double)f; }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    if (s == 123.456) out[0] = s;
}

int main()
{
    double *o; hipMalloc(&o, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    auto run = [&](const char *name, auto kern, int chains, int ops_per) {
        for (int wpc : {4, 8, 16, 32}) { // waves per CU
            const int blocks = 256 * wpc / 4;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, 10, 1.5); hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, iters, 1.5);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double ops = (double)blocks * 256 * iters * 8.0 * chains * ops_per;
            printf("%-22s chains %d waves/CU %2d : %7.2f T lane-ops/s\n", name, chains, wpc, ops / (ms * 1e-3) / 1e12);
        }
    };
    run("v_fma_f64", k_rate<0, 8>, 8, 1);
    run("v_fma_f64", k_rate<0, 2>, 2, 1);
    run("v_add_f64", k_rate<1, 8>, 8, 1);
    run("v_mul_f64", k_rate<2, 8>, 8, 1);
    run("v_rcp_f64", k_rate<3, 8>, 8, 1);
    run("cvt i32->f64 (+shift)", k_rate<4, 8>, 8, 1);
    run("f64->f32, fma32, ->f64", k_rate<5, 8>, 8, 1);
    return 0;
}
