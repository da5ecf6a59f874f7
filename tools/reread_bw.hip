// reread_bw.hip -- does the second pass of a two-pass row kernel (k_partition_stats: mean, then squared deviations)
// come out of the 256 MB Infinity Cache when few enough rows are in flight?  One workgroup per row of 200 000
// floats reads the row twice; the footprint in flight is (workgroups per CU) x 256 CUs x 800 KB.
//   hipcc --offload-arch=gfx950 -O3 -o reread_bw tools/reread_bw.hip && ./reread_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

// REV: the second pass walks the row from its END (what pass 1 read last is what a cache still holds)
template <int THREADS, int UNROLL, bool NT2, bool REV = false>
__global__ void __launch_bounds__(THREADS) k_twice(const float *__restrict__ x, int m, int T, int passes, float *out, int lds_pad)
{
    extern __shared__ float pad[]; // occupancy limiter
    const f4 *row = reinterpret_cast<const f4 *>(x + (size_t)blockIdx.x * m);
    f4 acc = {0, 0, 0, 0};
    const int nrounds = (T / 4) / (UNROLL * THREADS);
    for (int p = 0; p < passes; p++) {
        for (int rd = 0; rd < nrounds; rd++) {
            const int rr = (REV && p == passes - 1) ? nrounds - 1 - rd : rd;
            const int i = rr * UNROLL * THREADS + threadIdx.x;
            f4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
                v[u] = (NT2 && p == passes - 1) ? __builtin_nontemporal_load(&row[i + u * THREADS]) : row[i + u * THREADS];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc += v[u];
        }
        __syncthreads();
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) { out[0] = s; pad[lds_pad] = s; }
}

__global__ void k_fill(float *x, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 65536ull * 256ull) { unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; x[i] = 60.0f + (float)(h & 0xffff) * 0.001f; }
}

int main(int argc, char **)
{
    const int m = 201500, T = 200000;
    const size_t bytes = (size_t)24 << 30;
    const int rows = (int)(bytes / 4 / m);
    float *d, *o;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 0, bytes);
    if (argc > 1) { // non-zero, non-uniform contents (a zero-filled buffer could flatter a cache or the memory controller)
        hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, d, bytes / 4);
        (void)hipDeviceSynchronize();
    }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto run = [&](const char *name, auto kern, int threads, int per_cu, int passes) {
        const int lds = 160 * 1024 / per_cu - 512; // so that exactly per_cu workgroups fit a CU
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(rows), dim3(threads), lds, 0, d, m, T, passes, o, 0); (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(rows), dim3(threads), lds, 0, d, m, T, passes, o, 0);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-14s threads %4d x %d per CU (%4.0f MB in flight) passes %d : %6.2f ms  %6.2f TB/s of requested bytes\n", name, threads,
               per_cu, per_cu * 256 * 0.8, passes, ms, (double)passes * rows * T * 4.0 / (ms * 1e-3) / 1e12);
    };
    for (int passes : {1, 2}) {
        run("256", k_twice<256, 4, false>, 256, 5, passes);
        run("256", k_twice<256, 4, false>, 256, 2, passes);
        run("256", k_twice<256, 4, false>, 256, 1, passes);
        run("512", k_twice<512, 4, false>, 512, 2, passes);
        run("512", k_twice<512, 4, false>, 512, 1, passes);
        run("1024", k_twice<1024, 4, false>, 1024, 1, passes);
        run("1024 u2", k_twice<1024, 2, false>, 1024, 1, passes);
        run("1024 nt2", k_twice<1024, 4, true>, 1024, 1, passes);
        run("512 nt2", k_twice<512, 4, true>, 512, 2, passes);
        run("256 rev", k_twice<256, 4, false, true>, 256, 5, passes);
        run("256 rev nt2", k_twice<256, 4, true, true>, 256, 5, passes);
        run("256 rev", k_twice<256, 4, false, true>, 256, 2, passes);
        run("1024 rev", k_twice<1024, 4, false, true>, 1024, 1, passes);
    }
    return 0;
}
