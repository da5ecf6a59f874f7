// Where do the waves of ONE workgroup land?  (hipcc --offload-arch=gfx950 -O2 tools/wave_simd_placement.hip -o /tmp/wsp && /tmp/wsp)
// Every wave reads HW_REG_HW_ID (gfx9: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se [15:13]) and runs a
// dependent float chain; printed per launch shape: the SIMD of each wave, and the time of the chain with 1 .. 8 waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(int *out, float *sink, int steps, int lds_probe)
{
    extern __shared__ float lds[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = (int)id;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < steps; i++) { a = a * b + 0.5f; a = a < 0 ? 0 : a; }
    if (lds_probe) lds[threadIdx.x] = a;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main()
{
    int *out; float *sink;
    hipMalloc(&out, 4096 * 16 * 4); hipMalloc(&sink, 4096 * 512 * 4);
    int h[16];
    for (int lds : {0, 128 * 1024}) {
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int waves = 1; waves <= 8; waves++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), lds, 0, out, sink, 1000, lds ? 1 : 0);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), lds, 0, out, sink, 2000000, lds ? 1 : 0);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
            printf("lds %6d B, %d waves: chain of 2e6 steps %.2f ms (%.1f cycles/step at 2.4 GHz); simd of wave:", lds, waves, ms, ms * 1e-3 * 2.4e9 / 2e6);
            for (int w = 0; w < waves; w++) printf(" %d", (h[w] >> 4) & 3);
            printf("  cu:"); for (int w = 0; w < waves; w++) printf(" %d", (h[w] >> 8) & 15);
            printf("\n");
        }
    }
    return 0;
}
