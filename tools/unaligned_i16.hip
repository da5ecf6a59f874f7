// Can gfx950 fetch 8 bytes (four int16 samples) from an address that is only 2-byte aligned?  (int16-native streaming
// would read segments that start at arbitrary sample offsets.)  Build: hipcc --offload-arch=gfx950 -O3 tools/unaligned_i16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short s4u __attribute__((ext_vector_type(4), aligned(2)));
__global__ void k(const short *p, int off, int n, long long *out)
{
    long long s = 0;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 3 < n; i += gridDim.x * blockDim.x * 4) {
        const s4u v = *reinterpret_cast<const s4u *>(p + off + i);
        s += (long long)v.x + 3 * (long long)v.y + 5 * (long long)v.z + 7 * (long long)v.w;
    }
    atomicAdd((unsigned long long *)out, (unsigned long long)s);
}
int main()
{
    const int n = 1 << 24;
    std::vector<short> h(n + 8);
    for (int i = 0; i < n + 8; i++) h[i] = (short)((i * 2654435761u) >> 17);
    short *d; long long *o;
    hipMalloc(&d, (n + 8) * 2); hipMalloc(&o, 8);
    hipMemcpy(d, h.data(), (n + 8) * 2, hipMemcpyHostToDevice);
    for (int off = 0; off < 4; off++) {
        hipMemset(o, 0, 8);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, off, n, o);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long got; hipMemcpy(&got, o, 8, hipMemcpyDeviceToHost);
        long long want = 0;
        for (int i = 0; i + 3 < n; i += 4) want += (long long)h[off + i] + 3ll * h[off + i + 1] + 5ll * h[off + i + 2] + 7ll * h[off + i + 3];
        printf("offset %d samples: %s  %.1f GB/s\n", off, got == 20 * want ? "OK" : "MISMATCH", 20.0 * n * 2 / ms / 1e6);
    }
    return 0;
}
