// Texture-addresser (TA) rate microbenchmark for gfx950: cycles a CU's vector-memory path spends per wave64 gather instruction,
// as a function of load width and of how the 64 lane addresses fall into aligned 64-byte / 128-byte blocks. All data is L1-resident
// (each workgroup reads its own 8 KiB window), so this is the address/data-return path, not the cache hierarchy.
// Output: ns and shader cycles per wave-instruction per CU (4 SIMDs share one TA), at W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

struct Res { unsigned long long t, r; };

// PATTERN -> byte offset of lane l within the window (window 8 KiB)
__device__ __forceinline__ unsigned lane_offset(int pat, unsigned l, unsigned it) {
    switch (pat) {
    case 0: return l * 16;                               // contiguous, 64-B aligned quads
    case 1: return l * 16 + 16;                          // contiguous, every quad straddles a 64-B boundary
    case 2: return l * 16 + 32;                          // straddles at the half
    case 3: return l * 32;                               // stride 2 texels
    case 4: return ((l * 2654435761u + it * 40503u) >> 7) % 512 * 16; // random 16-B texels in the window
    case 5: return (l & 15) * 16 + (l >> 4) * 1024;      // 16x4 pixel wave footprint: 4 rows of 16 contiguous texels, aligned
    case 6: return (l & 15) * 16 + (l >> 4) * 1024 + 16; // same, shifted by one texel
    case 7: return (l & 15) * 16 + (l >> 4) * 1024 + 48; // same, shifted by three texels
    case 8: return l * 8;                                // 8-B contiguous
    case 9: return l * 8 + 8;                            // 8-B contiguous shifted
    case 10: return l * 4;                               // 4-B contiguous
    case 11: return ((l >> 1) & 7) * 32 + (l & 1) * 16 + ((l >> 4) * 2) * 1024; // 2x2-quad rotated style: pairs
    case 12: return (l & 3) * 16 + ((l >> 2) * 331 % 120) * 64;              // quads aligned to 64 B, quads scattered
    case 13: return (l & 3) * 16 + 16 + ((l >> 2) * 331 % 120) * 64;         // quads scattered, each straddling
    default: return 0;
    }
}

template <int WIDTH, bool BUFFER>
__global__ __launch_bounds__(256) void k_ta(const unsigned char* base, float* out, Res* res, int iters, int pat) {
    const unsigned l = threadIdx.x & 63;
    const unsigned char* win = base + (size_t)blockIdx.x * 8192; // SGPR base
    unsigned long long r0, r1;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r0));
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned acc = 0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)win, 0, 0xffffffff, 0x00020000);
    typedef unsigned int u4v __attribute__((__vector_size__(16)));
    typedef unsigned int u2v __attribute__((__vector_size__(8)));
    for (int it = 0; it < iters; ++it) {
        unsigned off = lane_offset(pat, l, (unsigned)it) & 8191u & ~(unsigned)(WIDTH * 4 - 1);
        if (WIDTH == 4) {
            u4v v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (BUFFER) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[k]) : "v"(off), "s"(rs));
                else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[k]) : "v"(off), "s"(win));
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k][0];
        } else if (WIDTH == 2) {
            u2v v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (BUFFER) asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(v[k]) : "v"(off), "s"(rs));
                else asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v[k]) : "v"(off), "s"(win));
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k][0];
        } else {
            unsigned v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (BUFFER) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v[k]) : "v"(off), "s"(rs));
                else asm volatile("global_load_dword %0, %1, %2" : "=v"(v[k]) : "v"(off), "s"(win));
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k];
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r1));
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
    if (l == 0) res[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Res{t1 - t0, r1 - r0};
}

typedef void (*kern_t)(const unsigned char*, float*, Res*, int, int);

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    const int iters = 1024;
    unsigned char* buf; float* out; Res* res;
    (void)hipMalloc(&buf, (size_t)cus * 8 * 8192 + 65536);
    (void)hipMemset(buf, 1, (size_t)cus * 8 * 8192 + 65536);
    (void)hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    (void)hipMalloc(&res, sizeof(Res) * cus * 8 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* pats[] = {"contig aligned x16B", "contig +16B", "contig +32B", "stride 32B", "random 16B", "16x4 rows aligned", "16x4 rows +16B", "16x4 rows +48B",
                          "contig 8B", "contig 8B +8", "contig 4B", "pairs (2x2 quads)", "quads aligned scattered", "quads straddling scattered"};
    struct K { const char* n; kern_t k; } ks[] = {{"global x4", k_ta<4, false>}, {"global x2", k_ta<2, false>}, {"global x1", k_ta<1, false>},
                                                   {"buffer x4", k_ta<4, true>}, {"buffer x2", k_ta<2, true>}, {"buffer x1", k_ta<1, true>}};
    printf("# %d CUs; 8 loads per iteration x %d iterations per wave; W waves per SIMD\n", cus, iters);
    printf("%-10s %-28s %2s %12s %14s\n", "load", "pattern", "W", "ns/inst/CU", "cyc/inst/CU");
    for (auto& k : ks)
        for (int pat = 0; pat < 14; ++pat)
            for (int W : {1, 4}) {
                int blocks = cus * W;
                for (int rep = 0; rep < 2; ++rep) {
                    (void)hipEventRecord(e0);
                    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, buf, out, res, iters, pat);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                }
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                std::vector<Res> h(blocks * 4);
                (void)hipMemcpy(h.data(), res, h.size() * sizeof(Res), hipMemcpyDeviceToHost);
                std::vector<double> t;
                for (auto& x : h) t.push_back((double)x.t);
                std::sort(t.begin(), t.end());
                double n_inst_cu = (double)iters * 8 * 4 * W; // wave-instructions per CU
                printf("%-10s %-28s %2d %12.3f %14.2f\n", k.n, pats[pat], W, ms * 1e6 / n_inst_cu, t[t.size() / 2] / n_inst_cu);
            }
    return 0;
}
