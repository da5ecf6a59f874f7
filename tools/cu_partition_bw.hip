// cu_partition_bw.hip -- can a streaming kernel and a float64-ALU kernel share MI355X profitably?
// (1) HBM read bandwidth of a pure streaming read as a function of the CUs it may use (hipExtStreamCreateWithCUMask) and of
//     the waves per SIMD it is allowed (launch bounds via dynamic LDS padding): is bandwidth a per-CU / per-wave-slot resource?
// (2) the same read beside an FMA-bound float64 kernel: unmasked on two streams, and with the CUs split between them.
//   hipcc --offload-arch=gfx950 -O3 -o cu_partition_bw tools/cu_partition_bw.hip && ./cu_partition_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void __launch_bounds__(256) k_read(const f4 *__restrict__ x, size_t n4, float *out)
{
    extern __shared__ float pad[]; // (dynamic LDS only limits the blocks per CU)
    f4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
    for (; i + (UNROLL - 1) * 256 < n4; i += stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = __builtin_nontemporal_load(&x[i + u * 256]);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) { out[0] = s; pad[threadIdx.x] = s; }
}

// the same read with `WORK` dependent-free float32 operations per sample (what a real streaming pass does beside loading)
template <int WORK>
__global__ void __launch_bounds__(256) k_read_work(const f4 *__restrict__ x, size_t n4, float *out, float t0, float t1)
{
    extern __shared__ float pad[];
    f4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x * 8 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
    for (; i + 7 * 256 < n4; i += stride) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(&x[i + u * 256]);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            f4 w = v[u];
#pragma unroll
            for (int k = 0; k < WORK / 2; k++) { w = w * t0 + t1; }   // 2 ops per sample and step
            acc += w;
        }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) { out[0] = s; pad[threadIdx.x] = s; }
}

// FMA-bound float64: 8 independent chains per lane, `iters` x 8 fmas; grid-stride over `n` work items of fixed cost
__global__ void __launch_bounds__(256) k_alu(double *out, int iters, int items)
{
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        const double m = 1.0000001, c = 1e-9;
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
            a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
        }
        const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        if (s == 1.2345) out[0] = s;
    }
}

static hipStream_t masked_stream(int n_cu_total, int first, int count, bool interleave)
{
    // bit i of the mask = CU i of the device's enumeration; interleave: every k-th CU instead of a contiguous range
    std::vector<uint32_t> mask((n_cu_total + 31) / 32, 0u);
    if (!interleave) for (int i = first; i < first + count && i < n_cu_total; i++) mask[i / 32] |= 1u << (i % 32);
    else {
        // `count` CUs spread evenly starting at offset `first`
        for (int j = 0; j < count; j++) { int i = (int)(((long long)j * n_cu_total) / count + first) % n_cu_total; mask[i / 32] |= 1u << (i % 32); }
    }
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return nullptr; }
    return s;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int NCU = pr.multiProcessorCount;
    printf("device: %s, %d CUs\n", pr.name, NCU);
    const size_t bytes = (size_t)16 << 30;
    float *d, *o; double *od;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMalloc(&od, 8);
    hipMemset(d, 0, bytes);
    const size_t n4 = bytes / 16;
    hipEvent_t a, b, c2, d2; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c2); hipEventCreate(&d2);
    auto time_read = [&](hipStream_t st, int grid, int unroll, size_t lds) {
        auto launch = [&] {
            if (unroll == 4) hipLaunchKernelGGL((k_read<4>), dim3(grid), dim3(256), lds, st, (const f4 *)d, n4, o);
            else hipLaunchKernelGGL((k_read<8>), dim3(grid), dim3(256), lds, st, (const f4 *)d, n4, o);
        };
        launch(); hipStreamSynchronize(st);
        hipEventRecord(a, st); for (int r = 0; r < 3; r++) launch(); hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        return 3.0 * bytes / (ms * 1e-3) / 1e12;
    };
    hipFuncSetAttribute((const void *)k_read<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k_read<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // (1a) bandwidth against blocks per CU (waves per SIMD) on the whole device
    printf("\n(1a) streaming read, all CUs, limited blocks per CU (256 threads = 1 wave per SIMD each):\n");
    for (int bpc : {1, 2, 3, 4, 6, 8}) {
        const size_t lds = (size_t)(160 * 1024 / bpc) - 1024;
        printf("  %d block(s)/CU: u4 %5.2f TB/s   u8 %5.2f TB/s\n", bpc, time_read(nullptr, 16384, 4, lds), time_read(nullptr, 16384, 8, lds));
    }
    // (2) the read beside a float64 FMA kernel
    const int iters = 4096, items = 8192 * 8;
    auto time_pair = [&](hipStream_t sr, hipStream_t sa, const char *label) {
        // alone
        hipLaunchKernelGGL((k_read<8>), dim3(16384), dim3(256), 0, sr, (const f4 *)d, n4, o); hipStreamSynchronize(sr);
        hipLaunchKernelGGL(k_alu, dim3(8192), dim3(256), 0, sa, od, iters, items); hipStreamSynchronize(sa);
        float ms_r, ms_a, ms_both;
        hipEventRecord(a, sr); for (int r = 0; r < 4; r++) hipLaunchKernelGGL((k_read<8>), dim3(16384), dim3(256), 0, sr, (const f4 *)d, n4, o); hipEventRecord(b, sr); hipEventSynchronize(b);
        hipEventElapsedTime(&ms_r, a, b);
        hipEventRecord(a, sa); hipLaunchKernelGGL(k_alu, dim3(8192), dim3(256), 0, sa, od, iters, items); hipEventRecord(b, sa); hipEventSynchronize(b);
        hipEventElapsedTime(&ms_a, a, b);
        // together: wall time from the first start to the last end
        hipDeviceSynchronize();
        hipEventRecord(a, sr); hipEventRecord(c2, sa);
        for (int r = 0; r < 4; r++) hipLaunchKernelGGL((k_read<8>), dim3(16384), dim3(256), 0, sr, (const f4 *)d, n4, o);
        hipLaunchKernelGGL(k_alu, dim3(8192), dim3(256), 0, sa, od, iters, items);
        hipEventRecord(b, sr); hipEventRecord(d2, sa);
        hipEventSynchronize(b); hipEventSynchronize(d2);
        float r_alone_in_pair, a_in_pair;
        hipEventElapsedTime(&r_alone_in_pair, a, b); hipEventElapsedTime(&a_in_pair, c2, d2);
        ms_both = r_alone_in_pair > a_in_pair ? r_alone_in_pair : a_in_pair;
        printf("  %-34s read alone %7.2f ms (%4.2f TB/s)  alu alone %7.2f ms  | together: read %7.2f  alu %7.2f  wall %7.2f ms = %4.2f of serial\n",
               label, ms_r, 4.0 * bytes / (ms_r * 1e-3) / 1e12, ms_a, r_alone_in_pair, a_in_pair, ms_both, ms_both / (ms_r + ms_a));
    };
    printf("\n(2) streaming read (4 x 16 GiB) beside a float64 FMA kernel:\n");
    { hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2); time_pair(s1, s2, "two plain streams (shared CUs)"); hipStreamDestroy(s1); hipStreamDestroy(s2); }
    // (1b) bandwidth against the number of CUs
    printf("\n(1b) streaming read (u8, 8 blocks/CU) on a CU-masked stream:\n");
    for (int inter = 0; inter < (getenv("CU_MASKS") ? 1 : 0); inter++)
        for (int n : {32, 64, 96, 128, 160, 192, 224, 256}) {
            if (n > NCU) continue;
            hipStream_t st = masked_stream(NCU, 0, n, inter != 0);
            if (!st) return 1;
            printf("  %3d CUs (%s): %5.2f TB/s\n", n, inter ? "spread" : "contiguous", time_read(st, 16384, 8, 0));
            hipStreamDestroy(st);
        }
    // (3) a PERSISTENT low-occupancy read (1 or 2 blocks per CU, resident before the ALU kernel starts) beside the ALU kernel
    printf("\n(3) persistent read with few blocks per CU, launched first, beside the float64 FMA kernel (two plain streams):\n");
    {
        hipStream_t sr, sa; hipStreamCreate(&sr); hipStreamCreate(&sa);
        auto pair = [&](const char *label, auto launch_read) {
            float ms_r, ms_a, t_r, t_a;
            launch_read(); hipStreamSynchronize(sr);
            hipEventRecord(a, sr); launch_read(); hipEventRecord(b, sr); hipEventSynchronize(b); hipEventElapsedTime(&ms_r, a, b);
            hipEventRecord(a, sa); hipLaunchKernelGGL(k_alu, dim3(8192), dim3(256), 0, sa, od, iters, items); hipEventRecord(b, sa); hipEventSynchronize(b);
            hipEventElapsedTime(&ms_a, a, b);
            hipDeviceSynchronize();
            hipEventRecord(a, sr); launch_read(); hipEventRecord(b, sr);
            hipEventRecord(c2, sa); hipLaunchKernelGGL(k_alu, dim3(8192), dim3(256), 0, sa, od, iters, items); hipEventRecord(d2, sa);
            hipEventSynchronize(b); hipEventSynchronize(d2);
            hipEventElapsedTime(&t_r, a, b); hipEventElapsedTime(&t_a, c2, d2);
            const float wall = t_r > t_a ? t_r : t_a;
            printf("  %-44s read alone %7.2f ms  alu alone %7.2f ms | together: read %7.2f  alu %7.2f  wall %7.2f = %4.2f of serial\n",
                   label, ms_r, ms_a, t_r, t_a, wall, wall / (ms_r + ms_a));
        };
        const int REP = 8; // 8 x 16 GiB per launch sequence ~ 19 ms at 7 TB/s: the ALU kernel's length
        for (int bpc : {1, 2, 8}) {
            char lb[64];
            snprintf(lb, 64, "pure read, %d block(s)/CU persistent", bpc);
            pair(lb, [&] { for (int r = 0; r < REP; r++) hipLaunchKernelGGL((k_read<8>), dim3(NCU * bpc), dim3(256), 0, sr, (const f4 *)d, n4, o); });
            snprintf(lb, 64, "read + 8 flop/sample, %d block(s)/CU", bpc);
            pair(lb, [&] { for (int r = 0; r < REP; r++) hipLaunchKernelGGL((k_read_work<8>), dim3(NCU * bpc), dim3(256), 0, sr, (const f4 *)d, n4, o, 1.0001f, 0.5f); });
            snprintf(lb, 64, "read + 16 flop/sample, %d block(s)/CU", bpc);
            pair(lb, [&] { for (int r = 0; r < REP; r++) hipLaunchKernelGGL((k_read_work<16>), dim3(NCU * bpc), dim3(256), 0, sr, (const f4 *)d, n4, o, 1.0001f, 0.5f); });
        }
        hipStreamDestroy(sr); hipStreamDestroy(sa);
    }
    if (getenv("CU_SPLIT"))
    for (int nr : {96, 128}) {
        hipStream_t sr = masked_stream(NCU, 0, nr, false), sa = masked_stream(NCU, nr, NCU - nr, false);
        char lb[64]; snprintf(lb, 64, "CU split %d read / %d alu", nr, NCU - nr);
        time_pair(sr, sa, lb);
        hipStreamDestroy(sr); hipStreamDestroy(sa);
    }
    return 0;
}
