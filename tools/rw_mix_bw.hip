// HBM bandwidth of plain read / write / copy streams, 16 bytes per lane, contiguous 1 KB per wave instruction (MI355X):
//   hipcc --offload-arch=gfx950 -O3 tools/rw_mix_bw.hip -o /tmp/rwmix && /tmp/rwmix
// What the split conv stack's layers do between their matrix work: every 64 -> 64 layer reads 1.8 MB and writes 1.8 MB per read.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_read(const f4 *a, f4 *sink, size_t n) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { f4 v = __builtin_nontemporal_load(a + i); acc += v; }
    if (acc.x == 12345.f) sink[0] = acc;
}
__global__ void k_write(f4 *a, size_t n) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = v;
}
__global__ void k_copy(const f4 *a, f4 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = __builtin_nontemporal_load(a + i);
}
// chunked: a block moves whole 68 KB chunks (the conv layers' tiles), blocks stride over the chunks
__global__ void k_copy_chunks(const f4 *a, f4 *b, size_t nchunks, int chunk16) {
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x)
        for (int i = threadIdx.x; i < chunk16; i += blockDim.x) b[c * chunk16 + i] = __builtin_nontemporal_load(a + c * chunk16 + i);
}
__global__ void k_write_chunks(f4 *b, size_t nchunks, int chunk16) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x)
        for (int i = threadIdx.x; i < chunk16; i += blockDim.x) b[c * chunk16 + i] = v;
}
int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    f4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, double moved, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-58s %6.2f ms  %5.2f TB/s\n", name, ms, moved / ms * 1e-9);
    };
    for (int blocks : {2048, 8192}) {
        printf("grid %d x 256\n", blocks);
        run("read 8 GB", bytes, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, b, n); });
        run("write 8 GB", bytes, [&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n); });
        run("copy 8 GB -> 8 GB (bytes moved = 16 GB)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    }
    const int chunk16 = 69632 / 16; // 68 KB
    const size_t nch = bytes / 69632;
    for (int blocks : {256, 512, 1024, 2048}) {
        char nm[96];
        snprintf(nm, sizeof nm, "copy in 68 KB chunks, %d persistent blocks x 256", blocks);
        run(nm, 2.0 * nch * 69632, [&] { hipLaunchKernelGGL(k_copy_chunks, dim3(blocks), dim3(256), 0, 0, a, b, nch, chunk16); });
        snprintf(nm, sizeof nm, "write in 68 KB chunks, %d persistent blocks x 256", blocks);
        run(nm, 1.0 * nch * 69632, [&] { hipLaunchKernelGGL(k_write_chunks, dim3(blocks), dim3(256), 0, 0, b, nch, chunk16); });
    }
    hipMemsetAsync(b, 0, bytes, 0); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; r++) hipMemsetAsync(b, 0, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %6.2f ms  %5.2f TB/s\n", "hipMemsetAsync 8 GB", ms / 5, bytes / (ms / 5) * 1e-9);
    return 0;
}
