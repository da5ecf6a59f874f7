// What does a burst of 16 ds_read_b128 / ds_write_b128 cost one wave when its lanes are rows of an LDS matrix (lane stride = row pitch)?
//   hipcc --offload-arch=gfx950 -O2 tools/lds_stride_probe.hip -o /tmp/lsp && /tmp/lsp
// One workgroup of 1 or 6 waves (every wave runs the same bursts on its own rows); per pitch (in floats): cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define LDS __attribute__((address_space(3)))
__global__ void k(float *sink, long long *stamps, int pitch, int lanes, int base_floats, int iters, int write)
{
    extern __shared__ float lds_raw[];
    LDS float *lds = (LDS float *)lds_raw;
    const int ln = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40000; i += blockDim.x) lds[i] = i;
    __syncthreads();
    LDS float *row = lds + base_floats + (ln % lanes) * pitch + wave * 64;
    v4f acc = {0, 0, 0, 0};
    long long c0, c1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    if (ln < lanes)
        for (int it = 0; it < iters; it++) {
            if (!write) {
                v4f t[16];
#pragma unroll
                for (int j = 0; j < 16; j++) t[j] = *(LDS v4f *)(row + 4 * j);
#pragma unroll
                for (int j = 0; j < 16; j++) acc += t[j];
            } else {
#pragma unroll
                for (int j = 0; j < 16; j++) *(LDS v4f *)(row + 4 * j) = acc + (float)j;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    if (threadIdx.x == 0) stamps[0] = c1 - c0;
    sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main()
{
    float *sink; long long *stamps;
    (void)hipMalloc(&sink, 4096 * 4); (void)hipMalloc(&stamps, 64);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000;
    for (int base : {0, 16384, 24576, 30000})
    for (int waves : {1, 6})
        for (int write : {0, 1})
            for (int lanes : {48})
                for (int pitch : {4, 68, 260, 64}) {
                    if (base + (lanes - 1) * pitch + 6 * 64 + 64 > 39000) continue;
                    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 159 * 1024, 0, sink, stamps, pitch, lanes, base, iters, write); (void)hipDeviceSynchronize(); }
                    long long h; (void)hipMemcpy(&h, stamps, 8, hipMemcpyDeviceToHost);
                    printf("base %5d floats, %d waves, %s, %2d lanes, pitch %3d floats: %6.1f cycles per ds_%s_b128\n", base, waves, write ? "write" : "read ", lanes, pitch, (double)h / iters / 16, write ? "write" : "read");
                }
    return 0;
}
