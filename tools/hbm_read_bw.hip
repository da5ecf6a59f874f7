// hbm_read_bw.hip -- what a kernel that does nothing but read can pull from HBM on this device: the practical ceiling
// the streaming passes of the detect path are compared with in DESIGN.md section 5 (6.3-6.5 TB/s with plain float4
// loads, 6.8 TB/s with non-temporal ones; flat and one-block-per-row layouts).
//   hipcc --offload-arch=gfx950 -O3 -o hbm_read_bw tools/hbm_read_bw.hip && ./hbm_read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_read(const f4 *__restrict__ x, size_t n4, float *out)
{
    f4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
    for (; i + (UNROLL - 1) * 256 < n4; i += stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(&x[i + u * 256]) : x[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}
// row-wise: block per row (like the detect kernels): rows of m floats, T used
template <int UNROLL>
__global__ void __launch_bounds__(256) k_rows(const float *__restrict__ x, int m, int T, int rows_per_block, float *out)
{
    f4 acc = {0, 0, 0, 0};
    for (int rr = 0; rr < rows_per_block; rr++) {
        const f4 *row = reinterpret_cast<const f4 *>(x + (size_t)(blockIdx.x * rows_per_block + rr) * m);
        int i = threadIdx.x;
        for (; i + (UNROLL - 1) * 256 < T / 4; i += UNROLL * 256) {
            f4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) v[u] = row[i + u * 256];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc += v[u];
        }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}
int main()
{
    const size_t bytes = (size_t)24 << 30;
    float *d, *o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4);
    hipMemset(d, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 3; r++) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-34s %7.2f TB/s\n", name, 3.0 * bytes / (ms * 1e-3) / 1e12);
    };
    const size_t n4 = bytes / 16;
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "flat u4 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_read<4, false>), dim3(g), dim3(256), 0, 0, (const f4 *)d, n4, o); });
        snprintf(nm, 64, "flat u8 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_read<8, false>), dim3(g), dim3(256), 0, 0, (const f4 *)d, n4, o); });
        snprintf(nm, 64, "flat u4 nt grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_read<4, true>), dim3(g), dim3(256), 0, 0, (const f4 *)d, n4, o); });
    }
    const int m = 201500, T = 200000;
    const int rows = (int)(bytes / 4 / m);
    for (int rpb : {1, 8}) {
        char nm[64];
        snprintf(nm, 64, "rows u2 rpb %d", rpb); run(nm, [&] { hipLaunchKernelGGL((k_rows<2>), dim3(rows / rpb), dim3(256), 0, 0, d, m, T, rpb, o); });
        snprintf(nm, 64, "rows u4 rpb %d", rpb); run(nm, [&] { hipLaunchKernelGGL((k_rows<4>), dim3(rows / rpb), dim3(256), 0, 0, d, m, T, rpb, o); });
        snprintf(nm, 64, "rows u8 rpb %d", rpb); run(nm, [&] { hipLaunchKernelGGL((k_rows<8>), dim3(rows / rpb), dim3(256), 0, 0, d, m, T, rpb, o); });
    }
    return 0;
}
