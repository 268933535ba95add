// TEST INFRASTRUCTURE (tests/test_sanitizer.py): proves that AddressSanitizer still sees through the fibers of the host emulation
// (tests/hip_emu/hip/hip_runtime.h: one ucontext per HIP thread, stack switches announced to ASan). A 16x16 block stages values in
// "LDS", passes a barrier - from here on every HIP thread runs on its own fiber stack - and then, with argv[1] == "fault", one thread
// writes one element past a heap plane. Expected: exit 0 without the argument, an ASan heap-buffer-overflow report with it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

__global__ void k_probe(float* plane, int n, int fault) {
    __shared__ float tile[256];
    const int tid = (int)threadIdx.y * 16 + (int)threadIdx.x;
    tile[tid] = (float)tid;
    __syncthreads();
    float v = tile[255 - tid];
    if (__syncthreads_or(v < 0.0f))
        return;
    plane[tid] = v;
    if (fault && tid == 200)
        plane[n] = v; // one past the end, written from a fiber stack
}

int main(int argc, char** argv) {
    const int n = 256;
    float* plane = nullptr;
    if (hipMalloc((void**)&plane, n * sizeof(float)) != hipSuccess)
        return 2;
    const int fault = argc > 1 && std::strcmp(argv[1], "fault") == 0;
    hipLaunchKernelGGL(k_probe, dim3(2, 1, 1), dim3(16, 16, 1), 0, nullptr, plane, n, fault);
    float sum = 0.0f;
    for (int i = 0; i < n; i++)
        sum += plane[i];
    std::printf("sum %.0f\n", sum);
    (void)hipFree(plane);
    return sum == 32640.0f ? 0 : 3;
}
