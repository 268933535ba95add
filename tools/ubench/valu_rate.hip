// VALU issue-rate microbenchmark, second pass: a wide list of the instructions the denoiser kernels are made of, at the
// occupancy the spatial passes run at (4 waves / SIMD), plus mixes (do "fast" and "slow" classes overlap?).
// Each row: wall ns per wave-instruction per SIMD (HIP events), shader cycles per instruction (s_memtime), effective clock from
// s_memtime / s_memrealtime (100 MHz constant counter) inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

struct Res { unsigned long long t, r; };

#define KERNEL(NAME, BODY)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(float* out, Res* res, int iters, float seed) {                  \
        float a[16]; float2 p[16];                                                                              \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = make_float2(a[i], a[i] + 0.5f); } \
        float b = seed * 0.5f + 1.0f, c = seed * 0.25f; float2 b2 = make_float2(b, b), c2 = make_float2(c, c); \
        unsigned u = threadIdx.x * 77u;                                                                         \
        unsigned long long r0, r1;                                                                              \
        asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r0));                                     \
        unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int it = 0; it < iters; ++it) { BODY }                                                             \
        unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r1));                                     \
        float s = (float)u;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + b2.x + c2.x;                                           \
        if ((threadIdx.x & 63) == 0) res[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Res{t1 - t0, r1 - r0}; \
    }

#define A1(TXT) asm volatile(TXT : "+v"(a[i_]) : "v"(b), "v"(c), "s"(seed));
#define DEF1(NAME, TXT) KERNEL(NAME, { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { asm volatile(TXT : "+v"(a[i_]) : "v"(b), "v"(c), "s"(seed) : "vcc"); } })
#define DEFP(NAME, TXT) KERNEL(NAME, { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { asm volatile(TXT : "+v"(p[i_]) : "v"(b2), "v"(c2)); } })
// two instructions per slot (mix): 16 slots = 32 instructions
#define DEF2(NAME, TXT) KERNEL(NAME, { _Pragma("unroll") for (int i_ = 0; i_ < 16; i_ += 2) { asm volatile(TXT : "+v"(a[i_]), "+v"(a[i_ + 1]) : "v"(b), "v"(c), "s"(seed) : "vcc"); } })

DEF1(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEF1(k_fmac, "v_fmac_f32 %0, %1, %2")
DEF1(k_fma_s, "v_fma_f32 %0, %0, %3, %2")
DEF1(k_fma_abs, "v_fma_f32 %0, |%0|, %1, %2 clamp")
DEF1(k_mul, "v_mul_f32 %0, %0, %1")
DEF1(k_mul_e64, "v_mul_f32_e64 %0, |%0|, %1 clamp")
DEF1(k_add, "v_add_f32 %0, %0, %1")
DEF1(k_sub_abs, "v_sub_f32_e64 %0, 1.0, |%0| clamp")
DEF1(k_max, "v_max_f32 %0, %0, %1")
DEF1(k_min, "v_min_f32 %0, %0, %1")
DEF1(k_med3, "v_med3_f32 %0, %0, %1, %2")
DEF1(k_cnd_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
DEF1(k_cnd_s, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
DEF1(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1")
DEF1(k_cmp_s, "v_cmp_lt_f32_e64 s[22:23], %0, %1")
DEF1(k_cmp_cls, "v_cmp_class_f32 vcc, %0, %1")
DEF1(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %0")
DEF1(k_cvt_f32_f16_sdwa, "v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
DEF1(k_cvt_f16_f32, "v_cvt_f16_f32 %0, %0")
DEF1(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
DEF1(k_cvt_i32, "v_cvt_i32_f32 %0, %0")
DEF1(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
DEF1(k_cvt_ubyte, "v_cvt_f32_ubyte0 %0, %0")
DEF1(k_floor, "v_floor_f32 %0, %0")
DEF1(k_fract, "v_fract_f32 %0, %0")
DEF1(k_rcp, "v_rcp_f32 %0, %0")
DEF1(k_rsq, "v_rsq_f32 %0, %0")
DEF1(k_exp, "v_exp_f32 %0, %0")
DEF1(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF1(k_mul24, "v_mul_u32_u24 %0, %0, %1")
DEF1(k_addu, "v_add_u32 %0, %0, %1")
DEF1(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEF1(k_lshl_add, "v_lshl_add_u32 %0, %0, 4, %1")
DEF1(k_lshl, "v_lshlrev_b32 %0, 4, %0")
DEF1(k_and, "v_and_b32 %0, %0, %1")
DEF1(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF1(k_bfe, "v_bfe_u32 %0, %0, 4, 8")
DEF1(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF1(k_mov, "v_mov_b32 %0, %1")
DEF1(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF1(k_add_dpp, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF1(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF1(k_fma_mix, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]")
DEF1(k_fma_mixlo, "v_fma_mixlo_f16 %0, %0, %1, %2")
DEF1(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
DEF1(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
DEF1(k_fma_f16, "v_fma_f16 %0, %0, %1, %2")
DEF1(k_dot2, "v_dot2_f32_f16 %0, %0, %1, %2")
DEF1(k_dot2c, "v_dot2c_f32_f16 %0, %1, %2")
DEFP(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
DEFP(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
DEFP(k_pk_add, "v_pk_add_f32 %0, %0, %1")
DEFP(k_pk_mov, "v_pk_mov_b32 %0, %1, %2")
// mixes: one fast + one slow per slot
DEF2(k_mix_fma_med3, "v_fma_f32 %0, %0, %2, %3\n v_med3_f32 %1, %1, %2, %3")
DEF2(k_mix_fma_cvt, "v_fma_f32 %0, %0, %2, %3\n v_cvt_f32_f16 %1, %1")
DEF2(k_mix_fma_mad24, "v_fma_f32 %0, %0, %2, %3\n v_mad_u32_u24 %1, %1, %2, %3")
DEF2(k_mix_fma_mul, "v_fma_f32 %0, %0, %2, %3\n v_mul_f32 %1, %1, %2")
DEF2(k_mix_fma_pkfma, "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3")
DEF2(k_mix_fma_salu, "v_fma_f32 %0, %0, %2, %3\n s_add_u32 s24, s24, 1\n v_fma_f32 %1, %1, %2, %3\n s_and_b64 s[26:27], s[26:27], s[22:23]")
DEF2(k_mix_dep, "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3")
DEF2(k_mix_dep_med3, "v_med3_f32 %0, %0, %2, %3\n v_med3_f32 %0, %0, %2, %3")

typedef void (*kern_t)(float*, Res*, int, float);
struct Row { const char* name; kern_t k; int per_slot; };

int main(int argc, char** argv) {
#define ROW(K) {#K, K, 1}
#define ROW2(K) {#K, K, 1}
    std::vector<Row> rows = {ROW(k_fma), ROW(k_fmac), ROW(k_fma_s), ROW(k_fma_abs), ROW(k_mul), ROW(k_mul_e64), ROW(k_add), ROW(k_sub_abs),
        ROW(k_max), ROW(k_min), ROW(k_med3), ROW(k_cnd_vcc), ROW(k_cnd_s), ROW(k_cmp_vcc), ROW(k_cmp_s), ROW(k_cmp_cls), ROW(k_cvt_f32_f16), ROW(k_cvt_f32_f16_sdwa),
        ROW(k_cvt_f16_f32), ROW(k_cvt_pkrtz), ROW(k_cvt_i32), ROW(k_cvt_f32_i32), ROW(k_cvt_ubyte), ROW(k_floor), ROW(k_fract), ROW(k_rcp), ROW(k_rsq), ROW(k_exp),
        ROW(k_mad24), ROW(k_mul24), ROW(k_addu), ROW(k_add3), ROW(k_lshl_add), ROW(k_lshl), ROW(k_and), ROW(k_and_or), ROW(k_bfe), ROW(k_perm), ROW(k_mov),
        ROW(k_mov_dpp), ROW(k_add_dpp), ROW(k_mul_lo), ROW(k_fma_mix), ROW(k_fma_mixlo), ROW(k_pk_fma_f16), ROW(k_pk_mul_f16), ROW(k_fma_f16), ROW(k_dot2), ROW(k_dot2c),
        ROW(k_pk_fma), ROW(k_pk_mul), ROW(k_pk_add), ROW(k_pk_mov),
        ROW(k_mix_fma_med3), ROW(k_mix_fma_cvt), ROW(k_mix_fma_mad24), ROW(k_mix_fma_mul), ROW(k_mix_fma_pkfma), ROW(k_mix_fma_salu), ROW(k_mix_dep), ROW(k_mix_dep_med3)};
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    const int iters = 2048;
    float* out; Res* res;
    (void)hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    (void)hipMalloc(&res, sizeof(Res) * cus * 8 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("# %d CUs; rows: 16 VALU instructions per loop iteration x %d iterations per wave; W waves per SIMD on every SIMD of the chip\n", cus, iters);
    printf("%-22s %3s %10s %10s %9s\n", "kernel", "W", "ns/inst", "cyc/inst", "clock GHz");
    int Ws[] = {1, 2, 4, 8};
    for (auto& r : rows)
        for (int W : Ws) {
            if (argc > 1 && W != atoi(argv[1])) continue;
            int blocks = cus * W;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, out, res, iters, 1.0f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<Res> h(blocks * 4);
            (void)hipMemcpy(h.data(), res, h.size() * sizeof(Res), hipMemcpyDeviceToHost);
            std::vector<double> t, rr;
            for (auto& x : h) { t.push_back((double)x.t); rr.push_back((double)x.r); }
            std::sort(t.begin(), t.end()); std::sort(rr.begin(), rr.end());
            double mt = t[t.size() / 2], mr = rr[rr.size() / 2];
            double n_inst = (double)iters * 16;
            printf("%-22s %3d %10.3f %10.3f %9.3f\n", r.name, W, ms * 1e6 / (n_inst * W), mt / (n_inst * W), mt / (mr * 10.0));
        }
    return 0;
}
