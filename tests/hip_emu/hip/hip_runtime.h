// TEST INFRASTRUCTURE: a tiny host-side stand-in for <hip/hip_runtime.h> so the UNMODIFIED product kernel sources
// (nrd-sample_amd/csrc/*.hip, nrdhip.cpp) can be compiled for the CPU with clang and checked against the oracle in the
// GPU-less authoring container (tests/test_kernels_emulated.py). It is never part of the product build: libnrdhip.so is
// always built by hipcc against the real HIP runtime, and this emulated library is only ever loaded by tests.
//
// Model: blocks run one after another; threads of a block run sequentially until one of them reaches a block-level
// primitive (__syncthreads*, __shfl_xor), at which point the block is re-run with one FIBER per HIP thread (ucontext, all on
// the calling OS thread: a barrier is "yield until every fiber has arrived", ~100 ns per switch instead of a mutex / condition
// variable hand-off between 256 OS threads; all kernels here are idempotent per block, so the re-run is safe). Under AddressSanitizer
// every stack switch is announced (fiber_switch); -DHIPEMU_USE_THREADS selects the older model: one OS thread per HIP thread, real barriers.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#ifndef HIPEMU_USE_THREADS
#include <ucontext.h>
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN_FIBERS 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__) && !defined(HIPEMU_ASAN_FIBERS)
#define HIPEMU_ASAN_FIBERS 1
#endif
#ifdef HIPEMU_ASAN_FIBERS
#include <sanitizer/common_interface_defs.h> // ASan must be told about every stack switch
#endif
#endif

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define NRD_WAVES_PER_EU(n)
// nrd_device.h: the scalar-path read of a per-tile texel (an aligned dword on the device) as the plain 1- / 2-byte read it stands for
#define NRD_TILE_TEXEL(addr, bytes) ((bytes) == 1 ? (uint32_t)*(const uint8_t*)(addr) : (uint32_t)*(const uint16_t*)(addr))
#define NRD_PIN_SGPRS8(a, b, c, d, e, f, g, h) ((void)0) // nrd_device.h: kernel-argument loads issued together
#define NRD_PIN_PLANES3(A, B, C) ((void)0)
#define NRD_PIN_PLANES4(A, B, C, D) ((void)0)
#define NRD_PIN_PLANES 1
#define NRD_RELOAD_ARGS(T, p, q) const T& q = (p) // nrd_device.h: the kernel arguments read afresh (scalar register pressure)
#define NRD_NO_DEVICE_ASM 1 // nrd_device.h: instruction-level forms (hdiff_) fall back to the plain expressions they equal
#define NRD_SCALAR_U32(ptr) (*(const uint32_t*)(ptr)) // nrd_device.h: a dword through the scalar data path
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 {
    uint32_t x, y;
};
struct uint4 {
    uint32_t x, y, z, w;
};
struct float4 {
    float x, y, z, w;
};
struct float2 {
    float x, y;
};

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
constexpr hipError_t hipSuccess = 0;
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
// everything runs synchronously on the host: streams and events order nothing
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// graphs: not emulated - capture reports "unsupported" and the host code takes its direct-launch path
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeThreadLocal = 1 };
enum hipGraphExecUpdateResult { hipGraphExecUpdateSuccess = 0 };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1, hipStreamCaptureStatusInvalidated = 2 };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return 1; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return 1; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphExecUpdate(hipGraphExec_t, hipGraph_t, hipGraphNode_t*, hipGraphExecUpdateResult*) { return 1; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return 1; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }

namespace hipemu {
struct Idx {
    unsigned x, y, z;
};
struct NeedThreads {};
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int count = 0, waiting = 0, generation = 0;
    void wait() {
        std::unique_lock<std::mutex> l(m);
        int g = generation;
        if (++waiting == count) {
            waiting = 0;
            generation++;
            cv.notify_all();
        } else
            cv.wait(l, [&] { return g != generation; });
    }
};
inline thread_local Idx t_threadIdx, t_blockIdx;
inline thread_local bool t_threaded = false;
inline Barrier g_barrier;
inline std::atomic<int> g_or{0};
inline float g_shfl[1024];
inline int g_blockThreads = 0;

#ifdef HIPEMU_USE_THREADS
template <typename F>
void run_block_lockstep(dim3 block, unsigned bx, unsigned by, unsigned bz, F& body) {
    int nthreads = (int)(block.x * block.y * block.z);
    g_barrier.count = nthreads;
    g_barrier.waiting = 0;
    g_blockThreads = nthreads;
    std::vector<std::thread> th;
    for (unsigned tz = 0; tz < block.z; tz++)
        for (unsigned ty = 0; ty < block.y; ty++)
            for (unsigned tx = 0; tx < block.x; tx++)
                th.emplace_back([=, &body] {
                    t_threaded = true;
                    t_blockIdx = {bx, by, bz};
                    t_threadIdx = {tx, ty, tz};
                    body();
                });
    for (auto& t : th)
        t.join();
}
inline void sync() {
    if (!t_threaded)
        throw NeedThreads();
    g_barrier.wait();
}
#else
// ---- fibers: every HIP thread of the block is a ucontext on the calling OS thread; sync() yields to the scheduler, which resumes the
// fibers round robin - one pass over all of them takes every fiber from barrier k to barrier k + 1 (or to its end)
struct Fibers {
    static constexpr size_t STACK = 256 * 1024;
    std::vector<ucontext_t> ctx;
    std::unique_ptr<char[]> stacks; // (not value-initialised: pages are touched as deep as a fiber really goes)
    size_t stackCount = 0;
    std::vector<char> done;
    ucontext_t sched;
    std::function<void()> body;
    int current = -1;
};
inline thread_local Fibers* t_fibers = nullptr;
inline thread_local Fibers t_fiberPool;
// stack switch with the AddressSanitizer bookkeeping around it (no-ops in the plain build)
inline void fiber_switch(ucontext_t* from, ucontext_t* to, const void* toStackBottom, size_t toStackSize, bool fromDies) {
#ifdef HIPEMU_ASAN_FIBERS
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(fromDies ? nullptr : &fake, toStackBottom, toStackSize);
    swapcontext(from, to);
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
    (void)toStackBottom;
    (void)toStackSize;
    (void)fromDies;
    swapcontext(from, to);
#endif
}
inline thread_local const void* t_schedStackBottom = nullptr;
inline thread_local size_t t_schedStackSize = 0;
inline void fiber_entry() {
#ifdef HIPEMU_ASAN_FIBERS
    __sanitizer_finish_switch_fiber(nullptr, &t_schedStackBottom, &t_schedStackSize); // first entry: learn the scheduler's stack
#endif
    Fibers& f = *t_fibers;
    f.body();
    f.done[f.current] = 1;
    fiber_switch(&f.ctx[f.current], &f.sched, t_schedStackBottom, t_schedStackSize, true); // never resumed
}
template <typename F>
void run_block_lockstep(dim3 block, unsigned bx, unsigned by, unsigned bz, F& body) {
    Fibers& f = t_fiberPool; // one pool per OS thread, shared by every kernel
    t_fibers = &f;
    const int n = (int)(block.x * block.y * block.z);
    g_blockThreads = n;
    if ((int)f.stackCount < n) {
        f.ctx.resize(n);
        f.stacks.reset(new char[(size_t)n * Fibers::STACK]);
        f.stackCount = (size_t)n;
    }
    f.done.assign(n, 0);
    f.body = [&body] { body(); };
    for (int i = 0; i < n; i++) {
        getcontext(&f.ctx[i]);
        f.ctx[i].uc_stack.ss_sp = f.stacks.get() + (size_t)i * Fibers::STACK;
        f.ctx[i].uc_stack.ss_size = Fibers::STACK;
        f.ctx[i].uc_link = &f.sched;
        makecontext(&f.ctx[i], fiber_entry, 0);
    }
    int live = n;
    while (live > 0) {
        for (int i = 0; i < n; i++) {
            if (f.done[i])
                continue;
            f.current = i;
            t_threaded = true;
            t_blockIdx = {bx, by, bz};
            t_threadIdx = {(unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y)};
            fiber_switch(&f.sched, &f.ctx[i], f.stacks.get() + (size_t)i * Fibers::STACK, Fibers::STACK, false);
            if (f.done[i])
                live--;
        }
    }
    t_threaded = false;
}
inline void sync() {
    if (!t_threaded)
        throw NeedThreads();
    Fibers& f = *t_fibers;
    fiber_switch(&f.ctx[f.current], &f.sched, t_schedStackBottom, t_schedStackSize, false);
}
#endif

// ---- optional gather trace (tools/gather_locality.py): for every wave (64 consecutive threads of a block, as on the GPU) and every
// buffer gather it executes - the k-th gather of each lane is the same instruction in these straight-line tap loops - the number of
// distinct 128-byte lines the 64 lanes touch. A hardware-independent measure of how well a pass's taps coalesce.
struct GatherTrace {
    bool on = false;
    std::vector<std::vector<uint64_t>> lanes;
    double instr = 0, lines = 0, laneLoads = 0;
};
inline GatherTrace g_gatherTrace;
inline thread_local int t_traceLane = -1;
inline void trace_gather(const void* a) {
    if (g_gatherTrace.on && t_traceLane >= 0)
        g_gatherTrace.lanes[(size_t)t_traceLane].push_back((uint64_t)(uintptr_t)a);
}
inline void trace_fold_block(int nthreads) {
    GatherTrace& g = g_gatherTrace;
    for (int w0 = 0; w0 < nthreads; w0 += 64) {
        size_t maxk = 0;
        for (int l = w0; l < w0 + 64 && l < nthreads; l++)
            maxk = std::max(maxk, g.lanes[(size_t)l].size());
        for (size_t k = 0; k < maxk; k++) {
            uint64_t seen[64];
            int ns = 0, nl = 0;
            for (int l = w0; l < w0 + 64 && l < nthreads; l++) {
                if (g.lanes[(size_t)l].size() <= k)
                    continue;
                nl++;
                const uint64_t line = g.lanes[(size_t)l][k] >> 7;
                bool dup = false;
                for (int q = 0; q < ns; q++)
                    dup = dup || seen[q] == line;
                if (!dup)
                    seen[ns++] = line;
            }
            g.instr += 1;
            g.lines += ns;
            g.laneLoads += nl;
        }
    }
}

template <typename F>
void launch(dim3 grid, dim3 block, F body) {
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                bool needThreads = false;
                t_threaded = false;
                t_blockIdx = {bx, by, bz};
                const int nthreads = (int)(block.x * block.y * block.z);
                const bool trace = g_gatherTrace.on;
                if (trace) {
                    g_gatherTrace.lanes.resize((size_t)nthreads);
                    for (auto& l : g_gatherTrace.lanes)
                        l.clear();
                }
                try {
                    for (unsigned tz = 0; tz < block.z && !needThreads; tz++)
                        for (unsigned ty = 0; ty < block.y; ty++)
                            for (unsigned tx = 0; tx < block.x; tx++) {
                                t_threadIdx = {tx, ty, tz};
                                t_traceLane = trace ? (int)((tz * block.y + ty) * block.x + tx) : -1;
                                body();
                            }
                } catch (NeedThreads&) {
                    needThreads = true;
                }
                t_traceLane = -1;
                if (needThreads)
                    run_block_lockstep(block, bx, by, bz, body); // (barrier kernels are not traced: none of them gathers through buffers)
                else if (trace)
                    trace_fold_block(nthreads);
            }
}
} // namespace hipemu

#define threadIdx (hipemu::t_threadIdx)
#define blockIdx (hipemu::t_blockIdx)

// NOTE: the emulated barrier requires every thread of the block to reach every barrier (true for the product kernels:
// block-uniform early exits happen before the first barrier, per-thread exits after the last one).
inline unsigned int __umul24(unsigned int a, unsigned int b) { return (a & 0xffffffu) * (b & 0xffffffu); }
// v_med3_f32: a NaN operand yields the minimum of the others
inline float hipemu_fmed3f(float a, float b, float c) {
    if (a != a) return b < c ? b : c;
    if (b != b) return a < c ? a : c;
    if (c != c) return a < b ? a : b;
    float lo = a < b ? a : b, hi = a < b ? b : a;
    float m = hi < c ? hi : c; // min(max(a, b), c)
    return lo > m ? lo : m;    // max(min(a, b), .)
}
#define __builtin_amdgcn_fmed3f hipemu_fmed3f
// raw buffer loads (V# = base pointer; stride / bounds unused by the product kernels)
struct __amdgpu_buffer_rsrc_t {
    const unsigned char* p;
};
inline __amdgpu_buffer_rsrc_t hipemu_make_rsrc(const void* p, int, unsigned, unsigned) { return {(const unsigned char*)p}; }
typedef unsigned int hipemu_u4v __attribute__((__vector_size__(16)));
typedef unsigned int hipemu_u2v __attribute__((__vector_size__(8)));
template <typename T>
inline T hipemu_buf_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    T v;
    hipemu::trace_gather(r.p + (unsigned)voff + (unsigned)soff);
    std::memcpy(&v, r.p + (unsigned)voff + (unsigned)soff, sizeof(T));
    return v;
}
#define __builtin_amdgcn_make_buffer_rsrc hipemu_make_rsrc
#define __builtin_amdgcn_raw_buffer_load_b128(r, v, s, a) hipemu_buf_ld<hipemu_u4v>(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b64(r, v, s, a) hipemu_buf_ld<hipemu_u2v>(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b32(r, v, s, a) hipemu_buf_ld<unsigned int>(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b16(r, v, s, a) hipemu_buf_ld<unsigned short>(r, v, s)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
// the transcendental instructions of the NRD_HW_TRANSCENDENTALS flavour (1 ULP on the device; here: the correctly rounded IEEE results)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_sqrtf(x) __builtin_sqrtf(x)
#define __builtin_amdgcn_exp2f(x) __builtin_exp2f(x)
// v_dot2_i32_i16: a.x * b.x + a.y * b.y + c (no clamp used)
typedef short hipemu_s2 __attribute__((ext_vector_type(2)));
static inline int hipemu_sdot2(hipemu_s2 a, hipemu_s2 b, int c) { return (int)a.x * (int)b.x + (int)a.y * (int)b.y + c; }
#define __builtin_amdgcn_sdot2(a, b, c, clamp) hipemu_sdot2(a, b, c)
// wave vote: threads run one at a time here, so a "wave" is the thread itself - valid wherever both sides of a wave-uniform
// branch compute the same result (the only way the product kernels use it)
#define __builtin_amdgcn_ballot_w64(pred) ((pred) ? 1ull : 0ull)
#define __builtin_amdgcn_readfirstlane(v) (v) // only used on values that are uniform by construction
inline void __syncthreads() { hipemu::sync(); }
inline int __syncthreads_or(int v) {
    hipemu::sync();
    if (v)
        hipemu::g_or.store(1);
    hipemu::sync();
    int r = hipemu::g_or.load();
    hipemu::sync();
    if (hipemu::t_threadIdx.x == 0 && hipemu::t_threadIdx.y == 0 && hipemu::t_threadIdx.z == 0)
        hipemu::g_or.store(0);
    hipemu::sync();
    return r;
}
inline float __shfl_xor(float v, int mask, int width) {
    (void)width;
    if (!hipemu::t_threaded)
        throw hipemu::NeedThreads();
    int tid = (int)(hipemu::t_threadIdx.y * 16 + hipemu::t_threadIdx.x); // product kernels use 16x16 blocks
    hipemu::g_shfl[tid] = v;
    hipemu::sync();
    float r = hipemu::g_shfl[(tid & ~63) | ((tid ^ mask) & 63)];
    hipemu::sync();
    return r;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch(grid, block, [&] { kernel(__VA_ARGS__); })

inline hipError_t hipMalloc(void** p, size_t n) {
    *p = std::malloc(n);
    return *p ? hipSuccess : 1;
}
inline hipError_t hipFree(void* p) {
    std::free(p);
    return hipSuccess;
}
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
    std::memcpy(dst, src, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {
    std::memcpy(dst, src, n);
    return hipSuccess;
}
inline hipError_t hipMemset(void* p, int v, size_t n) {
    std::memset(p, v, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    std::memset(p, v, n);
    return hipSuccess;
}
inline hipError_t hipMemset2DAsync(void* p, size_t pitch, int v, size_t w, size_t h, hipStream_t) {
    for (size_t y = 0; y < h; y++)
        std::memset((char*)p + y * pitch, v, w);
    return hipSuccess;
}
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
