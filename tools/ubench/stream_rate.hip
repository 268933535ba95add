// Streaming-rate microbenchmark: what a kernel that only moves bytes reaches on this GPU, in the access shapes the denoiser passes use.
//   linear : thread i copies 16 bytes at offset 16 i (one wave instruction = 1 KiB contiguous)
//   tile   : 16x16-pixel workgroups over a W x H plane of 16-byte texels, one texel per thread (a wave = 4 rows x 256 bytes), plain grid
//   read   : linear read only (sum into a register, one store per workgroup)    write: linear write only
// Prints GB/s (bytes read + bytes written per launch / mean launch time over the repeats).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_linear(const uint4* in, uint4* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_linear_nt(const uint4* in, uint4* out, size_t n) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load((const u4*)in + i), (u4*)out + i);
}
__global__ __launch_bounds__(256) void k_tile(const uint4* in, uint4* out, int W, int H) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x < W && y < H) out[(size_t)y * W + x] = in[(size_t)y * W + x];
}
__global__ __launch_bounds__(256) void k_read(const uint4* in, uint4* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 v = i < n ? in[i] : uint4{0, 0, 0, 0};
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = v; // never true: keeps the load
}
__global__ __launch_bounds__(256) void k_write(const uint4* in, uint4* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = uint4{(unsigned)i, 1u, 2u, 3u};
}
int main() {
    const int W = 3840, H = 2160;
    const size_t n = (size_t)W * H; // 16-byte texels: 133 MB per plane
    uint4 *a, *b;
    (void)hipMalloc(&a, n * 16 * 4); (void)hipMalloc(&b, n * 16 * 4);
    (void)hipMemset(a, 1, n * 16 * 4); (void)hipMemset(b, 2, n * 16 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 20;
    auto run = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 3; i++) launch(i);
        (void)hipEventRecord(e0);
        for (int i = 0; i < reps; i++) launch(i);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us per launch  %7.0f GB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
    };
    // every launch works on its own quarter of the 4-plane arenas, so that consecutive launches do not find their data in the 256 MiB Infinity Cache
    for (size_t planes : {(size_t)1, (size_t)2}) {
        size_t m = n * planes;
        unsigned blocks = (unsigned)((m + 255) / 256);
        char name[128];
        snprintf(name, sizeof name, "linear copy %zu MB (+ same out)", m * 16 >> 20);
        run(name, 2.0 * m * 16, [&](int i) { size_t off = (size_t)(i % (4 / planes)) * m; hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, a + off, b + off, m); });
        snprintf(name, sizeof name, "linear copy nt %zu MB", m * 16 >> 20);
        run(name, 2.0 * m * 16, [&](int i) { size_t off = (size_t)(i % (4 / planes)) * m; hipLaunchKernelGGL(k_linear_nt, dim3(blocks), dim3(256), 0, 0, a + off, b + off, m); });
        snprintf(name, sizeof name, "read only %zu MB", m * 16 >> 20);
        run(name, 1.0 * m * 16, [&](int i) { size_t off = (size_t)(i % (4 / planes)) * m; hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a + off, b, m); });
        snprintf(name, sizeof name, "write only %zu MB", m * 16 >> 20);
        run(name, 1.0 * m * 16, [&](int i) { size_t off = (size_t)(i % (4 / planes)) * m; hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, a, b + off, m); });
    }
    run("16x16 tile copy 133 MB (plain grid)", 2.0 * n * 16, [&](int i) { size_t off = (size_t)(i % 4) * n; hipLaunchKernelGGL(k_tile, dim3(W / 16, H / 16), dim3(256), 0, 0, a + off, b + off, W, H); });
    return 0;
}
