// nrd_device.h - device-side arithmetic and plane access of the MI355X NRD backend (gfx950, wave64).
//
// Numerics contract: every float expression is a sequence of separately rounded IEEE binary32 operations
// (-ffp-contract=off, IEEE divide/sqrt); fusion happens ONLY where the source says fma_() (v_fma_f32). That makes results
// reproducible bit for bit run to run, 1 GPU vs N GPUs, and against the CPU oracle of the tests. Transcendentals are fixed
// polynomials (no ocml calls on the pixel path). Instruction count matters, so the per-tap arithmetic avoids IEEE
// divisions and square roots: guides are pre-decoded once per pixel into a 16-byte texel, the normal weight works on the
// squared angle, the hit-distance weight has compact support, and taps are placed with the pixel-space Jacobian of the
// projection instead of a perspective divide per tap.
//
// Encodings (reference call sites): normal/roughness/materialID R10G10B10A2 pack Shaders/TraceOpaque.cs.hlsl:657;
// REBLUR hit distance normalisation Shaders/TraceOpaque.cs.hlsl:421; YCoCg radiance Shaders/TraceOpaque.cs.hlsl:756-757;
// GetSpecMagicCurve Shaders/Shared.hlsli:305-311; SIGMA penumbra/translucency Shaders/TraceOpaque.cs.hlsl:800-801.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#define NRD_DEV static __device__ __forceinline__
#define NRD_HD static __host__ __device__ __forceinline__
#ifndef NRD_WAVES_PER_EU // occupancy target of a kernel (the host emulation of the tests defines it away)
#define NRD_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif

// Projection flavour of a translation unit: the kernels are compiled twice from the same sources - perspective (default) and
// orthographic (nrd_reblur_ortho.hip / nrd_sigma_ortho.hip define NRD_ORTHO 1; the sample's "Ortho" camera, Source/NRDSample.cpp:1214,
// :1971). The flavour is a compile-time constant so that the perspective kernels carry no select for it; the host picks the launch
// namespace from FrameConsts::ortho.
#ifndef NRD_ORTHO
#define NRD_ORTHO 0
#endif
#if NRD_ORTHO
#define NRD_PROJ_NS ortho
#define NRD_KERNELS_BEGIN namespace ortho { namespace {
#define NRD_KERNELS_END } }
#else
#define NRD_PROJ_NS persp
#define NRD_KERNELS_BEGIN namespace {
#define NRD_KERNELS_END }
#endif

namespace nrdhip {

constexpr bool ORTHO = NRD_ORTHO != 0;

// ---- per-frame constants, identical layout on host and device -----------------------------------------------
struct FrameConsts {
    int W, H;         // rect of the whole (global) frame
    int Wprev, Hprev; // previous rect
    int resW, resH;   // local plane size
    int yOff;         // global row stored at local row 0 (row tiling)
    int ownY0, ownY1; // local rows this instance produces
    float invW, invH, invWprev, invHprev;
    float fr[4], frPrev[4]; // x0, y0, dx, dy
    float pv[4], pvPrev[4]; // view ray of pixel (px, gy): (pv0 + pv2 px, pv1 + pv3 gy, 1)
    float pj[5], pjPrev[5]; // m0, m5, m8, m9, s
    float w2v[9], w2vPrev[9], v2w[9], v2wPrev[9];
    float camDelta[3];
    float unproject, minRectDimMulUnproject;
    // pixel offset of a view-space tangent vector T scaled to one pixel of blur radius at the pixel's depth (kernel_basis_px below):
    // d(px) = +-jcx (T.x - rx T.z), d(py) = +-jcy (T.y - ry T.z) with (rx, ry, 1) the pixel's view ray; jcx = 0.5 W pj0 unproject / pj4,
    // jcy = -0.5 H pj1 unproject / pj4 (orthographic: no T.z term). The depth cancels: a world radius of `radius` pixels at depth z is
    // radius * unproject * |z|, and the perspective divide takes the |z| back
    float jcx, jcy;
    float denoisingRange, disocclusionThreshold, splitScreen;
    float disoccAlt; // CommonSettings::disocclusionThresholdAlternate, blended in per pixel by IN_DISOCCLUSION_THRESHOLD_MIX when ...
    int mixAvail;    // ... CommonSettings::isDisocclusionThresholdMixAvailable
    float mvScale[3];
    float viewZScale;
    uint32_t frameIndex;
    uint32_t strandMat;    // CommonSettings::strandMaterialID (Source/NRDSample.cpp:3871), 0xffffffff = none
    float strandThickness; // CommonSettings::strandThickness, world units
    uint32_t camAttachMat; // CommonSettings::cameraAttachedReflectionMaterialID (Source/NRDSample.cpp:3869-3876), 0xffffffff = none
    int mvWorld, confAvail, historyOk;
    int ortho; // orthographic projection: view position of pixel (px, gy) = (pv0 + pv2 px, pv1 + pv3 gy, z); pj = {m0, m5, m12, m13, 1}
    int tilesX, tilesY; // tile grid covering the owned rows: tile row 0 starts at local row tileY0 * 16
    int tileY0;
    int reverse; // 1: every XCD walks its tile sequence backwards (consecutive passes alternate: the reader starts where the writer ended)
    int prevY0, prevY1; // local rows [prevY0, prevY1) on which the PREVIOUS frame's planes are current (nrdhip_set_history_rows; whole planes
                        // on a single GPU). A row tiler refreshes only the rows reprojection may reach; a footprint texel outside is rejected
                        // like one outside the frame - a disocclusion - instead of being read from rows no exchange has written this frame
    // Tile order table of the launch grid (nrdhip.cpp tile_table, built once per tilesX x tilesY with xcd_tile_kj on the host): entry
    // j * 8 + k = the j-th tile of XCD k as tx | ty << 16 (ty relative to tileY0), 0xffffffff = spare workgroup. One scalar load replaces
    // the ~230 scalar / VALU instructions (five integer divisions) xcd_tile_kj costs every wave; nullptr = compute it (strips of a row
    // tiler whose shape has no table yet, launches recorded inside a graph capture before the table exists)
    const uint32_t* tileTable;
    // The per-tile flag of ClassifyTiles in the SAME order (one byte per table entry, written by this frame's ClassifyTiles through the
    // inverse table): a wave loads its table entry and its flag side by side - one scalar round trip - instead of entry, then flag at
    // (tx, ty). nullptr = read the Tiles plane (a launch whose grid is not the one ClassifyTiles ran on: strips of a row tiler)
    const uint8_t* tileFlags;
    int tilesPerXcd; // entries per XCD in tileTable = launch blocks / 8
    float rot[64][2];
};

struct PlaneRef {
    uint8_t* p;
    uint32_t pitch;
    uint16_t w, h; // only meaningful for planes sampled by uv (confidence)
};

// ---- small vector types ---------------------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};
struct f4 {
    float x, y, z, w;
};

NRD_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
NRD_HD float fmin2(float a, float b) { return a < b ? a : b; }
NRD_HD float fmax2(float a, float b) { return a > b ? a : b; }
NRD_HD float sat(float x) { return fmin2(fmax2(x, 0.0f), 1.0f); }
NRD_HD float clampf(float x, float a, float b) { return fmin2(fmax2(x, a), b); }
NRD_HD float lerpf(float a, float b, float t) { return fma_(b - a, t, a); }
NRD_DEV float smoothstep01(float x) {
    x = sat(x);
    return x * x * fma_(x, -2.0f, 3.0f); // == 3 - 2x rounded once: 2x is exact, so this is bit-identical to "3.0f - 2.0f * x"
}
NRD_DEV float absf(float x) { return __builtin_fabsf(x); } // a free source modifier (|x|) on the consuming instruction
NRD_HD int imin(int a, int b) { return a < b ? a : b; }
NRD_HD int imax(int a, int b) { return a > b ? a : b; }

NRD_DEV f3 add3(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
NRD_DEV f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
NRD_DEV f3 mul3(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
NRD_DEV float dot3(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
NRD_DEV f3 cross3(f3 a, f3 b) { return {fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))}; }
NRD_DEV f3 fma3(f3 a, float s, f3 c) { return {fma_(a.x, s, c.x), fma_(a.y, s, c.y), fma_(a.z, s, c.z)}; }
NRD_DEV f3 neg3(f3 a) { return {-a.x, -a.y, -a.z}; }
NRD_DEV f3 rot3(const float* m, f3 v) {
    return {fma_(m[2], v.z, fma_(m[1], v.y, m[0] * v.x)), fma_(m[5], v.z, fma_(m[4], v.y, m[3] * v.x)), fma_(m[8], v.z, fma_(m[7], v.y, m[6] * v.x))};
}
NRD_DEV f4 mul4(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
NRD_DEV f4 fma4(f4 a, float s, f4 c) { return {fma_(a.x, s, c.x), fma_(a.y, s, c.y), fma_(a.z, s, c.z), fma_(a.w, s, c.w)}; }
NRD_DEV f4 lerp4(f4 a, f4 b, float t) { return {lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t)}; }

NRD_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
NRD_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// ---- software reciprocal / reciprocal square root (numerics contract) ---------------------------------------------
// IEEE a / b and sqrt() expand to ~10 instructions around a quarter-rate v_rcp_f32 / v_sqrt_f32 on gfx950 (~13 issue slots); the
// per-pixel set-up of every pass is full of them. These are plain fma sequences (7 and 12 full-rate instructions) the CPU
// oracle repeats bit for bit: magic-constant seed + 3 Newton steps. rcp_ is correctly rounded for > 99.99 % of inputs
// (max 0.5 ulp), rsqrt_ is good to 1.3e-7 relative. Domain: positive, normal, finite x (the call sites guarantee it).
NRD_DEV float rcp_(float x) {
    float r = u2f(0x7EF311C7u - f2u(x));
    r = fma_(r, fma_(-x, r, 1.0f), r);
    r = fma_(r, fma_(-x, r, 1.0f), r);
    r = fma_(r, fma_(-x, r, 1.0f), r);
    return r;
}
NRD_DEV float rcps_(float x) { // any sign
    float r = rcp_(x < 0.0f ? -x : x);
    return x < 0.0f ? -r : r;
}
NRD_DEV float rsqrt_(float x) {
    float h = 0.5f * x;
    float r = u2f(0x5F3759DFu - (f2u(x) >> 1));
    r = r * fma_(-(h * r), r, 1.5f);
    r = r * fma_(-(h * r), r, 1.5f);
    r = r * fma_(-(h * r), r, 1.5f);
    return r;
}
NRD_DEV float sqrt_(float x) { return x * rsqrt_(x); } // sqrt_(0) = 0
// x^(1/3) for positive, normal, finite x (ledger row 19; the only consumer is TAA's XyzToLab, Shaders/Taa.cs.hlsl:43-54, whose
// pow(x, 0.333333) it stands in for): Newton on the INVERSE cube root - y <- y (4 - x y^3) / 3, division-free, three steps from a
// magic-constant seed (3.5 % -> 1.8e-3 -> 6.5e-6 -> rounding) - then x y^2. 19 instructions where exp2_poly(0.333333 log2_poly(x)) took
// ~50, eight times per pixel. Relative error 4.1e-7 against the cube root, <= 5e-6 against x^0.333333 on [0.008856, 1e6]
// (tests/test_oracle_math.py) - the shader's own pow is exp2(y log2 x) on 1-ULP hardware units, no closer than that.
NRD_DEV float cbrt_pos_(float x) {
    float y = u2f(0x54A21D2Au - f2u(x) / 3u);
    const float c = x * (1.0f / 3.0f);
    y = y * fma_(-c, (y * y) * y, 4.0f / 3.0f);
    y = y * fma_(-c, (y * y) * y, 4.0f / 3.0f);
    y = y * fma_(-c, (y * y) * y, 4.0f / 3.0f);
    return (x * y) * y;
}
NRD_DEV f3 normalize3(f3 a) {
    float l2 = dot3(a, a);
    float inv = rsqrt_(fmax2(l2, 1e-30f));
    return mul3(a, inv);
}

// ---- build flavour NRD_HW_TRANSCENDENTALS (libnrdhip_hwt.so) ----------------------------------------------------------------------
// The software sequences above and the polynomials below exist so that the CPU oracle can repeat every result bit for bit. What the
// reference's HLSL executes at those places are the GPU's transcendental instructions, and on gfx950 they are the cheaper choice:
// v_rcp_f32 / v_sqrt_f32 / v_exp_f32 issue in ~8 cycles per wave against 18 (rcp_), 31 (sqrt_), 13 (sqrt1_unscaled_) and 21
// (exp2_poly_neg) for the sequences (profiles/r02_ubench_valu_rate.txt), and the spatial passes are VALU-issue bound
// (profiles/r04v4_valu_issue_final_build.txt). This flavour uses the instructions (1 ULP each, more accurate than the one-step root
// and the degree-3 exponential they replace) in the WEIGHT-CLASS arithmetic of the spatial filters only: per-tap weights (normal chord,
// hit-distance / luminance exponential), their per-pixel parameters (1 / angle, 1 / plane distance scale, roughness and hit distance
// scales, sigma of A-trous) and the final 1 / weight sum. Everything a DISCRETE decision hangs on keeps the exact sequences: the
// projection Jacobian and blur radius (tap coordinates -> floor), reprojection, footprint validity, accumulation speeds (u8 codes),
// HistoryFix strides - so this flavour and its checker (liboracle_hwt.so: the same places with IEEE 1 / x, sqrtf, exp2f) gather the
// same texels and differ by rounding of weights only. Parity bar of the flavour: <= 1 ULP fp16 / PSNR >= 60 dB, not bit identity
// (tests/test_hw_transcendentals.py; measured distance: profiles/r05_ab_hw_transcendentals.txt).
#ifndef NRD_HW_TRANSCENDENTALS
#define NRD_HW_TRANSCENDENTALS 0
#endif
constexpr bool HW_TRANSCENDENTALS = NRD_HW_TRANSCENDENTALS != 0;
NRD_DEV float wrcp_(float x) { return HW_TRANSCENDENTALS ? __builtin_amdgcn_rcpf(x) : rcp_(x); }     // weight-class 1 / x, x > 0
NRD_DEV float wsqrt_(float x) { return HW_TRANSCENDENTALS ? __builtin_amdgcn_sqrtf(x) : sqrt_(x); } // weight-class sqrt(x), x >= 0

// ---- fp16 (hardware RNE converts, denormals on) ------------------------------------------------------------------
#define NRD_FP16_MAX 65504.0f
NRD_DEV float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
NRD_DEV uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)clampf(f, -NRD_FP16_MAX, NRD_FP16_MAX)); }
// b - a of two fp16 values that sit in the low / high halves of 32-bit words, rounded once to fp32: ONE v_fma_mix_f32 (b * 1.0 + (-a),
// the halves converted inside the instruction) where convert, convert, subtract takes three. The same single rounding as the float
// subtraction of the converted values - the host emulation of the tests and the oracle compute exactly that. (Written as
// fma_(h2f(b), 1.0f, -h2f(a)) the compiler folds the multiplication by one away and is back at three instructions.)
template <bool HI_B, bool HI_A>
NRD_DEV float hdiff_(uint32_t wb, uint32_t wa) {
#ifdef NRD_NO_DEVICE_ASM
    return h2f((uint16_t)(HI_B ? wb >> 16 : wb & 0xffffu)) - h2f((uint16_t)(HI_A ? wa >> 16 : wa & 0xffffu));
#else
    float r;
    if constexpr (!HI_B && !HI_A)
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(wb), "v"(wa));
    else if constexpr (HI_B && !HI_A)
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(wb), "v"(wa));
    else if constexpr (!HI_B && HI_A)
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(wb), "v"(wa));
    else
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(wb), "v"(wa));
    return r;
#endif
}
NRD_DEV f4 unpack_h4(uint2 v) { return {h2f((uint16_t)(v.x & 0xffffu)), h2f((uint16_t)(v.x >> 16)), h2f((uint16_t)(v.y & 0xffffu)), h2f((uint16_t)(v.y >> 16))}; }
// RGBA16_SNORM texel (the sample's DIRECTIONAL_OCCLUSION data format, Source/NRDSample.cpp:2937): v = max(int16 / 32767, -1)
NRD_DEV float sn2f(uint32_t h) { return fmax2((float)(int16_t)(uint16_t)h * (1.0f / 32767.0f), -1.0f); }
NRD_DEV uint32_t f2sn(float v) { return (uint32_t)(uint16_t)(int16_t)__builtin_floorf(fma_(fmin2(fmax2(v, -1.0f), 1.0f), 32767.0f, 0.5f)); }
NRD_DEV f4 unpack_sn4(uint2 v) { return {sn2f(v.x & 0xffffu), sn2f(v.x >> 16), sn2f(v.y & 0xffffu), sn2f(v.y >> 16)}; }
NRD_DEV uint2 pack_sn4(f4 v) { return {f2sn(v.x) | (f2sn(v.y) << 16), f2sn(v.z) | (f2sn(v.w) << 16)}; }
NRD_DEV uint2 pack_h4(f4 v) { return {(uint32_t)f2h(v.x) | ((uint32_t)f2h(v.y) << 16), (uint32_t)f2h(v.z) | ((uint32_t)f2h(v.w) << 16)}; }

// ---- polynomial transcendentals (coefficients frozen) ------------------------------------------------
NRD_HD float exp2_poly(float x) {
    x = clampf(x, -126.0f, 126.0f);
    float fi = __builtin_floorf(x + 0.5f);
    float f = x - fi;
    float p = 1.535336188319500e-4f;
    p = fma_(p, f, 1.339887440266574e-3f);
    p = fma_(p, f, 9.618437357674640e-3f);
    p = fma_(p, f, 5.550332471162809e-2f);
    p = fma_(p, f, 2.402264791363012e-1f);
    p = fma_(p, f, 6.931472028550421e-1f);
    p = fma_(p, f, 1.0f);
    int e = (int)fi;
    return p * u2f((uint32_t)(e + 127) << 23);
}

NRD_DEV float log2_poly(float x) {
    if (!(x > 1.17549435e-38f))
        return -126.0f;
    uint32_t u = f2u(x);
    int e = (int)((u >> 23) & 0xffu) - 127;
    float m = u2f((u & 0x7fffffu) | 0x3f800000u);
    if (m > 1.41421356f) {
        m = m * 0.5f;
        e += 1;
    }
    float t = m - 1.0f;
    float z = t * t;
    float p = 7.0376836292e-2f;
    p = fma_(p, t, -1.1514610310e-1f);
    p = fma_(p, t, 1.1676998740e-1f);
    p = fma_(p, t, -1.2420140846e-1f);
    p = fma_(p, t, 1.4249322787e-1f);
    p = fma_(p, t, -1.6668057665e-1f);
    p = fma_(p, t, 2.0000714765e-1f);
    p = fma_(p, t, -2.4999993993e-1f);
    p = fma_(p, t, 3.3333331174e-1f);
    float y = t * z * p;
    y = y - 0.5f * z;
    float ln = t + y;
    return ln * 1.44269504f + (float)e;
}

NRD_DEV float pow01(float x, float y) {
    x = sat(x);
    if (x <= 0.0f)
        return 0.0f;
    return exp2_poly(y * log2_poly(x));
}

NRD_DEV float atan_pos(float x) {
    bool inv = x > 1.0f;
    float t = inv ? wrcp_(x) : x; // (only consumer: spec_lobe_half_angle -> the normal weight's parameter)
    float s = t * t;
    float p = 0.0208351f;
    p = fma_(p, s, -0.0851330f);
    p = fma_(p, s, 0.1801410f);
    p = fma_(p, s, -0.3302995f);
    p = fma_(p, s, 0.9998660f);
    p = p * t;
    return inv ? 1.57079633f - p : p;
}

// NRD_UPSTREAM_FORMULAS = 1 (the DEFAULT since round 4: libnrdhip.so, liboracle.so): the RECALLED upstream forms of ledger rows 1, 2,
// 7, 13 (oracle/README.md) - hit-distance weight exp(-3 |x|), normal weight on the ANGLE (through an arccosine), Blur rotation per
// pixel, RELAX in linear RGB throughout (RELAX_LINEAR_RGB below). NRD_UPSTREAM_FORMULAS = 0 builds the cheaper forms rounds 1-3 had
// frozen (compact-support hit weight, squared-angle normal weight, rotation per 2x2 quad, YCoCg inside RELAX) as libnrdhip_frozen.so /
// liboracle_frozen.so: same sources, -DNRD_UPSTREAM_FORMULAS=0; bench.py times it beside the default (config.frozen_formulas) with the
// distance between the two outputs.
#ifndef NRD_UPSTREAM_FORMULAS
#define NRD_UPSTREAM_FORMULAS 1
#endif
constexpr bool UPSTREAM_FORMULAS = NRD_UPSTREAM_FORMULAS != 0;
constexpr int BLUR_ROTATION_SHIFT = UPSTREAM_FORMULAS ? 0 : 1; // Blur's Poisson rotation: per pixel (upstream) / per 2x2 quad (frozen)

// The angle between two normals as upstream's weights take it: Math::AcosApprox(cos) = sqrt(2) sqrt(saturate(1 - cos)) (MathLib, recalled:
// the library is not vendored in the sample) - the CHORD |n_a - n_b| of the two unit vectors, within 1 % of the arc below 28 degrees and
// 10 % short of it at 90. On guide normals the chord is (2 / 1023) sqrt(d2) with d2 the squared distance of the 10-bit codes
// (normal_dist2 below), so a tap pays one square root and nothing else: the A&S 4.4.45 arccosine polynomial the first default-flavour
// build evaluated here (6 more instructions per tap) was FARTHER from upstream than this.
// The square root: magic seed + ONE Newton step with tuned constants (J. Kadlec's 0x5F1FFFF9 / 0.703952253 / 2.38924456 variant of
// the classic sequence): relative error <= 6.5e-4 over every possible d2 (all 3 x 1023^2 integers checked, tests/test_oracle_math.py).
// That is as fine as its argument deserves - d2 counts quantisation steps of 2 / 1023, a chord of 0.02 (the narrowest lobe) is d2 ~ 100,
// known to +-5 % - and it runs once per tap of every spatial pass, which are priced in instructions (profiles/r04_valu_issue.txt): 6
// instructions where the two-step version took 11. Returns sqrt(x) / SQRT1_SCALE: the scale goes into the per-pixel weight parameter.
constexpr float SQRT1_SCALE = 0.703952253f;
NRD_DEV float sqrt1_unscaled_(float x) {
    const float y = u2f(0x5F1FFFF9u - (f2u(x) >> 1));
    const float u = x * y;
    return u * fma_(-u, y, 2.38924456f); // sqrt1_unscaled_(0) = 0
}
// chord of two guide normals = NORMAL_CHORD_SCALE * chord_unscaled_(d2); hwt flavour: v_sqrt_f32, no scale to undo
constexpr float NORMAL_CHORD_SCALE = (2.0f / 1023.0f) * (HW_TRANSCENDENTALS ? 1.0f : SQRT1_SCALE);
NRD_DEV float chord_unscaled_(float d2) { return HW_TRANSCENDENTALS ? __builtin_amdgcn_sqrtf(d2) : sqrt1_unscaled_(d2); }
// 2^x for x <= 0 (the hit-distance weight's exponent): round-to-nearest-even split (v_rndne_f32), a DEGREE-3 minimax polynomial on
// [-0.5, 0.5] (relative error 8.0e-5, a sixth of an fp16 ULP of the signals the weight multiplies; exp2_poly's degree 6 reaches 1.1e-7:
// three fma less per tap), the power of two applied by v_ldexp_f32
NRD_DEV float exp2_poly_neg(float x) {
    x = fmax2(x, -126.0f);
    const float fi = __builtin_rintf(x); // v_rndne_f32 (ties to even: f = +-0.5 is inside the fit either way)
    const float f = x - fi;
    float p = 5.519811809062958e-2f;
    p = fma_(p, f, 2.4267692863941193e-1f);
    p = fma_(p, f, 6.932618021965027e-1f);
    p = fma_(p, f, 9.999227523803711e-1f);
    return __builtin_ldexpf(p, (int)fi);
}
NRD_DEV float exp2_neg(float x) { return HW_TRANSCENDENTALS ? __builtin_amdgcn_exp2f(x) : exp2_poly_neg(x); } // hwt flavour: v_exp_f32
// hit-distance weight: compact-support stand-in for exp(-3|x|) (division-free): (1 - |x|)^2 clamped; upstream flavour: exp(-3 |x|)
constexpr float EXP_WEIGHT_SCALE = UPSTREAM_FORMULAS ? 4.32808512f : 1.0f; // 3 log2(e)
NRD_DEV float exp_weight(float ax) {
    if (UPSTREAM_FORMULAS)
        return exp2_neg(-EXP_WEIGHT_SCALE * ax); // ax >= 0
    float t = sat(1.0f - ax);
    return t * t;
}
// the same weight of |v| when the caller has EXP_WEIGHT_SCALE folded into v (the spatial passes' taps: v = fma(hitT, a, b) with per-pixel
// a, b - one multiply less per tap)
NRD_DEV float exp_weight_prescaled(float v) {
    if (UPSTREAM_FORMULAS)
        return exp2_neg(-absf(v));
    float t = sat(1.0f - absf(v));
    return t * t;
}
// Normal weight, from the squared distance d2 of two normals' 10-bit codes (normal_dist2 below; 1 - cos = d2 NORMAL_D2_TO_1MCOS).
// Frozen form: on the squared angle (angle^2 ~ 2 (1 - cos)), sqrt-free; its per-pixel parameter is w2 = 1 / angleMax^2 (normal_weight)
// or -2 w2 (normal_weight_m2: scaling by 2 is exact, so fma(-2 t, w2, 1) and fma(t, -2 w2, 1) round the same exact product - one
// multiply less per tap). Default flavour: smoothstep(1 - angle / angleMax) on the chord (sqrt1_unscaled_ above), parameter = 1 / angleMax for
// both; the chord's scale (NORMAL_CHORD_SCALE) is folded into the parameter (a per-pixel product the compiler hoists out of the tap loops).
constexpr float NORMAL_D2_TO_1MCOS = 0.5f * (2.0f / 1023.0f) * (2.0f / 1023.0f);
NRD_DEV float nw_param(float normalW) { return UPSTREAM_FORMULAS ? normalW : normalW * normalW; }
NRD_DEV float nw_param_m2(float normalW) { return UPSTREAM_FORMULAS ? normalW : -2.0f * (normalW * normalW); }
NRD_DEV float normal_weight(float d2, float prm) {
    if (UPSTREAM_FORMULAS)
        return smoothstep01(fma_(-chord_unscaled_(d2), prm * NORMAL_CHORD_SCALE, 1.0f));
    return smoothstep01(fma_(-2.0f * sat(1.0f - fma_(d2, -NORMAL_D2_TO_1MCOS, 1.0f)), prm, 1.0f));
}
NRD_DEV float normal_weight_m2(float d2, float prm) {
    if (UPSTREAM_FORMULAS)
        return smoothstep01(fma_(-chord_unscaled_(d2), prm * NORMAL_CHORD_SCALE, 1.0f));
    return smoothstep01(fma_(sat(1.0f - fma_(d2, -NORMAL_D2_TO_1MCOS, 1.0f)), prm, 1.0f));
}

// ---- input decode (once per pixel, in the ClassifyTiles passes) ---------------------------------------------------
NRD_DEV f3 oct_decode(float px, float py) {
    float fx = px * 2.0f - 1.0f, fy = py * 2.0f - 1.0f;
    float nz = 1.0f - absf(fx) - absf(fy);
    float t = sat(-nz);
    float nx = fx + (fx >= 0.0f ? -t : t);
    float ny = fy + (fy >= 0.0f ? -t : t);
    return normalize3({nx, ny, nz});
}

// v / 1023 for a 10-bit v, bit for bit the IEEE quotient the oracle computes with `/` (all 1024 inputs checked in exact arithmetic,
// tests/test_oracle_math.py): product with the rounded reciprocal + one fma residual correction - 3 full-rate instructions
// instead of the ~13 issue slots of the IEEE division expansion, three times per pixel in ClassifyTiles
NRD_DEV float unorm10_(uint32_t v) {
    const float x = (float)v, r = 1.0f / 1023.0f;
    const float q = x * r;
    return fma_(fma_(-q, 1023.0f, x), r, q);
}

// guide texel (8 bytes, round 3; == the guide part of the Blur / PostBlur tap texels):
//   .x = viewZ rounded to 22 bits | roughness as the 10-bit code of IN_NORMAL_ROUGHNESS. Read back AS ONE FLOAT it is the depth (the
//        roughness code perturbs it by < 2^-13 relative) - every consumer reads it that way
//   .y = normal x | y << 10 | z << 20, 10 bits per component (n = code * 2/1023 - 1, not re-normalised) | materialID << 30
// Written once per pixel by the ClassifyTiles passes. Round 3 measured what the passes pay for - bytes and memory instructions, not
// arithmetic (profiles/r03_ab_setup_planes.txt) - so the guide is as small as the input allows: the normal arrives in
// IN_NORMAL_ROUGHNESS as a 10 + 10 bit octahedron and the roughness as 10 bits, 3 x 10 bits and the original roughness code are as fine
// as that; only the depth loses its 10 low mantissa bits. A decode is a handful of bfe / cvt / fma instead of 3 fp16 converts.
constexpr int GUIDE_BYTES = 8;
// A pixel without geometry - |viewZ| beyond the denoising range, Inf, NaN (a viewZ plane filled with 0xFF bytes to say "no hit" is a NaN
// whose bit pattern would WRAP in the rounding below and come back as depth ~0) - stores ONE canonical depth: the largest finite float
// with its 10 low bits clear. Finite, so the arithmetic of a tap that lands on it stays free of Inf x 0, and beyond any range.
constexpr uint32_t GUIDE_SKY_DEPTH = 0x7F7FFC00u;
NRD_DEV uint32_t qn10(float v) { return (uint32_t)__builtin_floorf(clampf(fma_(v, 511.5f, 512.0f), 0.0f, 1023.0f)); }
// `geo`: the pixel has geometry - decided on the value that is STORED (the rounding may lift a depth within 2^-13 of the range over
// it), so that the ClassifyTiles passes flag tiles by the very test every consumer of the texel applies
NRD_DEV uint2 encode_guide(float z, uint32_t packedNR, float range, bool& geo) {
    f3 n = oct_decode(unorm10_(packedNR & 1023u), unorm10_((packedNR >> 10) & 1023u));
    const uint32_t code = (packedNR >> 20) & 1023u;
    const uint32_t w0 = ((f2u(z) + 0x200u) & 0xFFFFFC00u) | code;
    geo = absf(z) <= range && absf(u2f(w0)) <= range;
    return uint2{geo ? w0 : (GUIDE_SKY_DEPTH | code), qn10(n.x) | (qn10(n.y) << 10) | (qn10(n.z) << 20) | ((packedNR >> 30) << 30)};
}

struct Guide {
    float z;
    f3 n;
    float roughness;
    uint32_t mat;
    bool sky;
    uint32_t nw; // the texel's normal | material word (normal_cos works on the codes)
};

// Cosine of the angle between two guide normals, from their 10-bit CODES: cos = 1 - |n_a - n_b|^2 / 2 (exact for unit vectors). The
// dot product of two quantised, not re-normalised vectors is useless for what the normal weights need - 1 - cos at the 1e-4 level for
// the narrow specular lobes: |n|^2 is off by up to 2e-3, so identical normals would score like a 3 degree bend - while the squared
// difference of the codes is exact (integers below 2^24 in fp32), zero for equal normals and as fine as the quantisation step.
// Round 4 (profiles/r04_valu_issue.txt: the spatial passes sit on VALU issue, and the three bfe / sub / cvt chains of this distance were
// a quarter of a tap's cycles): the distance is formed in INTEGERS with packed 16-bit lanes - x and z of a normal word are one
// `v_and` away from being the two 16-bit lanes {x, z << 4}, so a tap costs and + bfe (y), v_pk_sub_i16, v_sub, v_pk_ashrrev_i16 (lane
// shifts {0, 4}: exact, 16 dz is a multiple of 16), v_mul_i32_i24 (dy^2), v_dot2c_i32_i16, v_cvt: 8 instructions instead of 12, and
// the same integer, hence the same float (|d|^2 <= 3 x 1023^2 < 2^24).
#ifndef NRD_NORMAL_DOT2
#define NRD_NORMAL_DOT2 1
#endif
typedef short nrd_s2 __attribute__((ext_vector_type(2)));
struct NormalCodes {
    nrd_s2 xz; // {x, z << 4}
    int y;
};
NRD_DEV NormalCodes normal_codes(uint32_t nw) { return {__builtin_bit_cast(nrd_s2, nw & 0x3ff003ffu), (int)((nw >> 10) & 1023u)}; }
NRD_DEV float normal_dist2(NormalCodes centre, uint32_t nw) {
    float d2;
    if (NRD_NORMAL_DOT2) {
        nrd_s2 d = __builtin_bit_cast(nrd_s2, nw & 0x3ff003ffu) - centre.xz; // {dx, 16 dz}, |16 dz| <= 16368
        d = d >> nrd_s2{0, 4};
        const int dy = (int)((nw >> 10) & 1023u) - centre.y;
        d2 = (float)__builtin_amdgcn_sdot2(d, d, dy * dy, false);
    } else { // the float form of rounds 1-3 (same value: every term an integer below 2^24)
        const float dx = (float)(nw & 1023u) - (float)(int)centre.xz.x, dy = (float)((nw >> 10) & 1023u) - (float)centre.y,
                    dz = (float)((nw >> 20) & 1023u) - (float)((int)centre.xz.y >> 4);
        d2 = fma_(dz, dz, fma_(dy, dy, dx * dx));
    }
    return d2;
}
NRD_DEV float normal_cos(NormalCodes centre, uint32_t nw) { return fma_(normal_dist2(centre, nw), -NORMAL_D2_TO_1MCOS, 1.0f); }

NRD_DEV Guide decode_guide(uint2 g, float range) {
    Guide r;
    r.nw = g.y;
    r.z = u2f(g.x);
    r.roughness = (float)(g.x & 1023u) * (1.0f / 1023.0f);
    const float s = 2.0f / 1023.0f;
    r.n = {fma_((float)(g.y & 1023u), s, -1.0f), fma_((float)((g.y >> 10) & 1023u), s, -1.0f), fma_((float)((g.y >> 20) & 1023u), s, -1.0f)};
    r.mat = g.y >> 30;
    r.sky = !(absf(r.z) <= range);
    return r;
}
NRD_DEV uint2 ld_guide(const PlaneRef& P, int x, int y);

// ---- tap texels of REBLUR's Blur / PostBlur (flavours without SH) --------------------------------------------------------------
// What a tap needs - depth, normal, roughness, material and the signal - in ONE 16-byte texel per signal, so a tap is one gather
// instead of two (guide + radiance): {guide texel (.x .y) | signal {Y, Co | Cg, hitT} as 4 x fp16 (.z .w)}. HistoryFix copies the
// pixel's guide texel in, Blur copies it through, and both passes take their CENTRE pixel's guide from the texel too (no guide plane
// access).
NRD_DEV Guide unpack_tap_guide(uint32_t w0, uint32_t w1, float range) { return decode_guide(uint2{w0, w1}, range); }

NRD_DEV f3 linear_to_ycocg(f3 c) {
    float Y = c.x * 0.25f + c.y * 0.5f + c.z * 0.25f;
    float Co = c.x * 0.5f - c.z * 0.5f;
    float Cg = c.y * 0.5f - c.x * 0.25f - c.z * 0.25f;
    return {Y, Co, Cg};
}
NRD_DEV f3 ycocg_to_linear(f3 c) {
    float t = c.x - c.z;
    return {fmax2(t + c.y, 0.0f), fmax2(c.x + c.z, 0.0f), fmax2(t - c.y, 0.0f)};
}
NRD_DEV f4 rgb_to_ycocg4(f4 v) {
    f3 c = linear_to_ycocg({v.x, v.y, v.z});
    return {c.x, c.y, c.z, v.w};
}

// Ledger row 13 (NRD_UPSTREAM_FORMULAS flavour): RELAX keeps its radiance in linear RGB from input to output, as upstream does; the
// frozen build converts to YCoCg in the PrePass and back in the last A-trous iteration (one code path with REBLUR). Wherever the
// RELAX passes want the luminance of a texel it is then Rec.709 of the rgb instead of channel 0, and a luminance clamp scales all
// three channels.
constexpr bool RELAX_LINEAR_RGB = UPSTREAM_FORMULAS;
NRD_DEV float luma709(f4 v) { return fma_(v.x, 0.2126f, fma_(v.y, 0.7152f, v.z * 0.0722f)); }
NRD_DEV float signal_luma(f4 v, bool relax) { return (RELAX_LINEAR_RGB && relax) ? luma709(v) : v.x; }
NRD_DEV float texel_luma(uint2 texel, bool relax) { return (RELAX_LINEAR_RGB && relax) ? luma709(unpack_h4(texel)) : h2f((uint16_t)texel.x); }
// luminance Y of `v` replaced by Yc (scale = Yc / Y), chroma (or the rgb ratios) kept
NRD_DEV void clamp_luma(f4& v, float Yc, float scale, bool relax) {
    v.x = (RELAX_LINEAR_RGB && relax) ? v.x * scale : Yc;
    v.y *= scale;
    v.z *= scale;
}

NRD_DEV float spec_magic_curve(float roughness) {
    float f = 1.0f - exp2_poly(-200.0f * roughness * roughness);
    return f * sqrt_(sat(roughness));
}

// roughness-dependent factor of the hit-distance normalisation; for the diffuse signal (roughness 1) it depends on the settings
// only and arrives precomputed from the host (ReblurParams::hitFactorDiff - same function, same roundings)
NRD_HD float reblur_hitdist_factor(const float* hp, float roughness) { return lerpf(1.0f, hp[2], exp2_poly(hp[3] * roughness * roughness)); }
NRD_DEV float reblur_hitdist_norm(float absViewZ, const float* hp, float roughness) {
    return fma_(absViewZ, hp[1], hp[0]) * reblur_hitdist_factor(hp, roughness);
}

NRD_DEV float spec_lobe_half_angle(float roughness) {
    float m = sat(roughness);
    m = m * m;
    return atan_pos(m * 3.0f);
}

NRD_DEV float spec_dominant_factor(float roughness) {
    float s = sat(1.0f - roughness);
    return s * (sqrt_(s) + roughness);
}

NRD_HD uint32_t hash_px(uint32_t x, uint32_t y, uint32_t frame, uint32_t salt) {
    uint32_t h = (x * 73856093u) ^ (y * 19349663u) ^ (frame * 83492791u) ^ (salt * 2654435761u);
    h ^= h >> 13;
    h *= 0x5bd1e995u;
    h ^= h >> 15;
    return h;
}

NRD_DEV void basis3(f3 n, f3& t, f3& b) {
    float sz = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = -rcps_(sz + n.z);
    float nxa = n.x * a, sx = sz * n.x;
    float bb = nxa * n.y;
    t = {fma_(sx, nxa, 1.0f), sz * bb, -sx};
    b = {bb, fma_(n.y * a, n.y, sz), -n.y};
}

// perspective: the view-space xy of a pixel scale with z; orthographic: they do not
NRD_DEV float zpersp(float z) { return ORTHO ? 1.0f : z; }
NRD_DEV f3 reconstruct(const float* fr, float u, float v, float z) { return {zpersp(z) * (u * fr[2] + fr[0]), zpersp(z) * (v * fr[3] + fr[1]), z}; }
NRD_DEV f3 reconstruct_px(const float* pv, float px, float gy, float z) { return {zpersp(z) * fma_(pv[2], px, pv[0]), zpersp(z) * fma_(pv[3], gy, pv[1]), z}; }
// unit vector from the view-space point toward the viewer
NRD_DEV f3 to_viewer(f3 Xv) {
    if (ORTHO)
        return {0.0f, 0.0f, Xv.z >= 0.0f ? -1.0f : 1.0f};
    return mul3(normalize3(Xv), -1.0f);
}
NRD_DEV bool project(const float* pj, f3 X, float& u, float& v) {
    if (ORTHO) {
        u = 0.5f + 0.5f * (pj[0] * X.x + pj[2]);
        v = 0.5f - 0.5f * (pj[1] * X.y + pj[3]);
        return true;
    }
    float cw = pj[4] * X.z;
    if (!(cw > 1e-6f))
        return false;
    float inv = rcp_(cw);
    u = 0.5f + 0.5f * ((pj[0] * X.x + pj[2] * X.z) * inv);
    v = 0.5f - 0.5f * ((pj[1] * X.y + pj[3] * X.z) * inv);
    return true;
}

NRD_DEV bool material_mismatch(uint32_t a, uint32_t b, uint32_t minMaterial) { return a != b && (a > b ? a : b) >= minMaterial; }
// The same predicate for a tap loop (one pixel against many): "a != b and one of them >= minMaterial" <=> class(a) != class(b) with
// class(m) = max(m, max(minMaterial, 1) - 1) - every material below the threshold falls into one class. The pixel's class and the
// floor are computed once; a tap pays shift + max + compare instead of shift + compare + max + compare (+ a scalar or).
NRD_DEV uint32_t material_floor(uint32_t minMaterial) { return (minMaterial > 1u ? minMaterial : 1u) - 1u; }
NRD_DEV uint32_t material_class(uint32_t m, uint32_t floor_) { return m > floor_ ? m : floor_; }

// per-pixel geometry of the bilateral passes: plane-distance weight is |zs * (ga0 + gax px + gay gy) + geoB|
// (orthographic: |zs * geoB + (ga0 + gax px + gay gy)| - geoB holds the z coefficient, ga0 absorbs the plane offset)
struct PixelGeo {
    f3 Xv, Nv;
    float rx, ry; // perspective: the pixel's view ray (rx, ry, 1); orthographic: its view-space xy
    float absZ, frustumSize;
    float ga0, gax, gay, geoB;
};
NRD_DEV PixelGeo pixel_geo(const FrameConsts& c, const Guide& g, int x, int gy, float planeDistSensitivity) {
    PixelGeo p;
    p.rx = fma_(c.pv[2], (float)x, c.pv[0]);
    p.ry = fma_(c.pv[3], (float)gy, c.pv[1]);
    p.Xv = {zpersp(g.z) * p.rx, zpersp(g.z) * p.ry, g.z}; // == reconstruct_px(c.pv, x, gy, g.z)
    p.Nv = rot3(c.w2v, g.n);
    p.absZ = absf(g.z);
    p.frustumSize = c.minRectDimMulUnproject * zpersp(p.absZ);
    float geoA = wrcp_(planeDistSensitivity * p.frustumSize);
    p.gax = p.Nv.x * c.pv[2] * geoA;
    p.gay = p.Nv.y * c.pv[3] * geoA;
    if (ORTHO) {
        p.ga0 = (fma_(p.Nv.x, c.pv[0], p.Nv.y * c.pv[1]) - dot3(p.Nv, p.Xv)) * geoA;
        p.geoB = p.Nv.z * geoA;
    } else {
        p.ga0 = fma_(p.Nv.x, c.pv[0], fma_(p.Nv.y, c.pv[1], p.Nv.z)) * geoA;
        p.geoB = -dot3(p.Nv, p.Xv) * geoA;
    }
    return p;
}
// Hair (CommonSettings::strandMaterialID / strandThickness, Source/NRDSample.cpp:3871-3872; the sample's own guide treatment:
// Shaders/TraceOpaque.cs.hlsl:644-649): per-pixel normals of strands thinner than a pixel are unreliable, so the normal-weight
// parameter of such pixels is scaled by lerp(0.25, 1, saturate(strandThickness / pixel world size))
NRD_DEV float strand_normal_relax(const FrameConsts& c, uint32_t mat, float absZ) {
    if (mat != c.strandMat)
        return 1.0f;
    return lerpf(0.25f, 1.0f, sat(c.strandThickness * wrcp_(c.unproject * zpersp(absZ))));
}
// Pixel offsets of a view-space tangent pair (T, B) of the blur kernel PER PIXEL OF BLUR RADIUS - the pixel-space Jacobian of the
// projection at the pixel, applied to the kernel basis scaled to the world size of one pixel at the pixel's depth (unproject |z|).
// Round 6: the perspective divide cancels that depth - with (rx, ry, 1) the view ray, d(px) / d(X.x) |z| unproject = sign(z) jcx and
// the z column is -rx times the x column - so the Jacobian is two frame constants, the ray and a sign: 10 instructions and no
// reciprocal where rounds 2-5 spent ~36 (rcp of clip w, the NDC position, four mul / fma / mul chains per signal). The taps of a pass
// are then J * radius * disk. j = {T -> px, T -> py, B -> px, B -> py}
NRD_DEV float signed_by(float k, float z) { return u2f(f2u(k) ^ (f2u(z) & 0x80000000u)); } // k * sign(z), exact
NRD_DEV void kernel_basis_px(const FrameConsts& c, float z, float rx, float ry, f3 T, f3 B, float (&j)[4]) {
    if (ORTHO) {
        j[0] = c.jcx * T.x;
        j[1] = c.jcy * T.y;
        j[2] = c.jcx * B.x;
        j[3] = c.jcy * B.y;
        return;
    }
    const float sx = signed_by(c.jcx, z), sy = signed_by(c.jcy, z);
    j[0] = sx * fma_(-rx, T.z, T.x);
    j[1] = sy * fma_(-ry, T.z, T.y);
    j[2] = sx * fma_(-rx, B.z, B.x);
    j[3] = sy * fma_(-ry, B.z, B.y);
}
// plane-distance term of a tap from its precomputed linear part ga = ga0 + gax px + gay gy
NRD_DEV float geo_plane(const PixelGeo& p, float ga, float zs) { return ORTHO ? fma_(zs, p.geoB, ga) : fma_(zs, ga, p.geoB); }
NRD_DEV float geo_weight(const PixelGeo& p, float px, float gy, float zs) {
    float ga = fma_(p.gax, px, fma_(p.gay, gy, p.ga0));
    return smoothstep01(1.0f - absf(geo_plane(p, ga, zs)));
}

// ---- plane access ------------------------------------------------------------------------------------------------
// Texel addressing: every plane is < 4 GiB (8K x 32 B = 1 GiB), so the byte offset is a 32-bit quantity built with one
// 24-bit multiply-add (rows < 2^24, pitch < 2^24) and the access becomes "global_load vdst, voffset, sbase" - no 64-bit
// (quarter-rate) address arithmetic per tap.
NRD_DEV uint32_t texel_offset(const PlaneRef& P, int x, int y, int bpt, int off) {
    return __umul24((uint32_t)y, P.pitch) + (uint32_t)x * (uint32_t)bpt + (uint32_t)off;
}
template <typename T>
NRD_DEV T ld(const PlaneRef& P, int x, int y, int bpt, int off = 0) {
    return *reinterpret_cast<const T*>(P.p + texel_offset(P, x, y, bpt, off));
}
template <typename T>
NRD_DEV void st(const PlaneRef& P, int x, int y, int bpt, T v, int off = 0) {
    *reinterpret_cast<T*>(P.p + texel_offset(P, x, y, bpt, off)) = v;
}
NRD_DEV uint2 ld_guide(const PlaneRef& P, int x, int y) { return ld<uint2>(P, x, y, GUIDE_BYTES); }
// Streaming accesses (the "nt" bit of the memory instruction): lines without reuse inside the kernel - an output nobody reads before
// the launch is over, a plane read at the thread's own pixel only - should not evict the lines the tap gathers live on. Measured per
// kernel (profiles/r03_ab_setup_planes.txt): a win where the data is not wanted again soon, a loss where the NEXT kernel reads it at
// once (the Infinity Cache carries planes from one pass to the next), so it is applied case by case.
template <typename T>
struct nt_native {
    typedef T type;
};
template <>
struct nt_native<uint2> {
    typedef unsigned int type __attribute__((ext_vector_type(2)));
};
template <>
struct nt_native<uint4> {
    typedef unsigned int type __attribute__((ext_vector_type(4)));
};
template <typename T>
NRD_DEV T ld_stream(const PlaneRef& P, int x, int y, int bpt, int off = 0) {
    typedef typename nt_native<T>::type N;
    return __builtin_bit_cast(T, __builtin_nontemporal_load(reinterpret_cast<const N*>(P.p + texel_offset(P, x, y, bpt, off))));
}
template <typename T>
NRD_DEV void st_stream(const PlaneRef& P, int x, int y, int bpt, T v, int off = 0) {
    typedef typename nt_native<T>::type N;
    __builtin_nontemporal_store(__builtin_bit_cast(N, v), reinterpret_cast<N*>(P.p + texel_offset(P, x, y, bpt, off)));
}

// Gathers of the spatial passes go through buffer instructions: a raw V# (base = the plane's first texel, no stride, no bounds -
// the tap positions are clamped before they become offsets) + a 32-bit byte offset per lane ("buffer_load_dwordx4 v, voff, s[V#],
// 0 offen"). Same texture-addresser cost as global_load (16 cycles per wave64 instruction of <= 16 B per lane when the lanes'
// texels are row-contiguous, ~26 when scattered: tools/ubench/ta_rate.hip), but the base lives in 4 SGPRs shared by all taps.
// `row0` = the global row held at local row 0 (row tiling): folded into the base once, so taps address GLOBAL rows
// (nrdhip.cpp denoise_parts checks that (first row + rows held) x pitch stays a 32-bit offset).
struct PlaneBuf {
    __amdgpu_buffer_rsrc_t r;
    uint32_t pitch;
};
NRD_DEV PlaneBuf plane_buf(const PlaneRef& P, int row0) {
    return {__builtin_amdgcn_make_buffer_rsrc(P.p - (size_t)row0 * P.pitch, 0, 0xffffffffu, 0x00020000), P.pitch}; // gfx9 raw-buffer word 3
}
template <typename T>
NRD_DEV T ldb(const PlaneBuf& B, int x, int y, int bpt, int off = 0) {
    typedef unsigned int u4v __attribute__((__vector_size__(16)));
    typedef unsigned int u2v __attribute__((__vector_size__(8)));
    const int o = (int)(__umul24((uint32_t)y, B.pitch) + (uint32_t)x * (uint32_t)bpt + (uint32_t)off);
    if constexpr (sizeof(T) == 16) {
        u4v v = __builtin_amdgcn_raw_buffer_load_b128(B.r, o, 0, 0);
        return T{v[0], v[1], v[2], v[3]};
    } else if constexpr (sizeof(T) == 8) {
        u2v v = __builtin_amdgcn_raw_buffer_load_b64(B.r, o, 0, 0);
        return T{v[0], v[1]};
    } else if constexpr (sizeof(T) == 4) {
        return (T)__builtin_amdgcn_raw_buffer_load_b32(B.r, o, 0, 0);
    } else {
        static_assert(sizeof(T) == 2, "ldb: 2 / 4 / 8 / 16-byte texels");
        return (T)__builtin_amdgcn_raw_buffer_load_b16(B.r, o, 0, 0);
    }
}

// 8-tap Poisson disk + weight (same frozen table as the oracle)
#define NRD_POISSON8_TABLE                                                                                          \
    {{-0.4706069f, -0.4427112f, 0.7592f}, {-0.9057375f, 0.3003471f, 0.5483f}, {-0.3487388f, 0.4037880f, 0.8287f},    \
     {0.1023042f, 0.6439373f, 0.7554f},   {0.5699277f, 0.3513750f, 0.7439f},  {0.2939128f, -0.1131226f, 0.9366f},    \
     {0.7836658f, -0.4208784f, 0.5932f},  {0.1564120f, -0.8198990f, 0.6314f}}
__device__ static const float g_poisson8[8][3] = NRD_POISSON8_TABLE;

// Passes that rotate the disk once per FRAME (all lanes share the rotation) get their 8 rotated offsets from the host in the
// kernel arguments (SGPRs) - the same two roundings per coordinate the per-pixel path performs, done once instead of per tap.
NRD_HD void rotate_taps(const float (*rot)[2], uint32_t frameIndex, uint32_t salt, const float (*disk)[3], float (*out)[2]) {
    uint32_t h = hash_px(0u, 0u, frameIndex, salt);
    float rc = rot[h & 63u][0], rs = rot[h & 63u][1];
    for (int t = 0; t < 8; t++) {
        out[t][0] = fma_(disk[t][0], rc, -(disk[t][1] * rs));
        out[t][1] = fma_(disk[t][0], rs, disk[t][1] * rc);
    }
}

// XCD-aware tile assignment with a cache-sized traversal. The dispatcher round-robins consecutive workgroups over the 8 XCDs
// (workgroup b runs on XCD b % 8), each XCD has its own 4 MiB L2, and every pass reads neighbours of its pixels (taps reach 8-58
// pixels, 5x5 stencils, motion-displaced footprints). So:
//  * XCD k owns a contiguous band of tile COLUMNS [k cols, (k + 1) cols), cols = ceil(tilesX / 8): neighbour reads stay in one L2;
//  * inside its band the XCD walks row-major, in column strips of at most ~NRD_STRIP_TARGET tiles (one strip at 4K, where a band
//    is 30 tiles wide; two at 8K): the ~128-256 workgroups an XCD has in flight then cover a compact 30 x 4..8 tile block instead of
//    half a tile row of the whole frame, and the rows above / below that the taps reach are still in L2 when the next rows run.
//  * the band an XCD works on ROTATES every ceil(tilesY / 8) tile rows (block (band, by) -> XCD (band + 3 by) mod 8): every XCD
//    visits every column band and every block row once, so content that is cheap in whole rows (sky above a horizon: those
//    tiles exit at once) or in whole columns costs every XCD the same. A fixed assignment of frame regions to XCDs does not: round
//    1's mapping (XCD k = the k-th run of full-width tile rows) left the XCDs that owned the sky rows of the bench scene (28 % of
//    the frame) idle while the others worked - CUs 70 % busy (SQ_BUSY_CU_CYCLES), now 92 %.
// Measured at 4K (profiles/r02_ab_tile_traversal.txt): round 1's mapping (XCD k = a run of full-width tile rows) 5870 Mpix/s;
// column bands 7060-7130 (strips of 5 / 6 / 8 / 10 / 15 / 30 tiles inside the band: 6870 / 6810 / 6780 / 7060 / 7050 / 7120). The
// three spatial passes and TemporalAccumulation gain 17-25 % each - what looked like a pure texture-addresser bound was to a good
// part the addresser stalling on L1 / L2 misses (TA_ADDR_STALLED_BY_TC).
NRD_HD int xcd_cols(int tilesX) { return (tilesX + 7) >> 3; }
#ifndef NRD_BAND_ROTATE // 1: the band an XCD works on rotates every xcd_rows() tile rows (load balance for any scene layout)
#define NRD_BAND_ROTATE 1
#endif
NRD_HD int xcd_rows(int tilesY) { return NRD_BAND_ROTATE ? (tilesY + 7) >> 3 : tilesY; } // tile rows per block row (<= 8 block rows)
NRD_HD int xcd_grid_blocks(int tilesX, int tilesY) { // launch size: 8 x (largest band x block rows, rounded up)
    const int rows = xcd_rows(tilesY);
    return xcd_cols(tilesX) * rows * ((tilesY + rows - 1) / rows) * 8;
}
NRD_HD bool xcd_tile_kj(const FrameConsts& c, const int k, const int jIn, int& tx, int& ty) { // j-th tile of XCD k
    const int cols = xcd_cols(c.tilesX), rows = xcd_rows(c.tilesY);
    const int perBlock = cols * rows;
    // c.reverse: the launch walks the XCD's tile sequence back to front. A pass that reads what the previous one wrote then starts in
    // what the 256 MiB Infinity Cache still holds of a 133-266 MB plane instead of chasing the eviction front (which pass runs reversed:
    // nrdhip.cpp directed(), measured per pass)
    const int j = c.reverse ? perBlock * ((c.tilesY + rows - 1) / rows) - 1 - jIn : jIn;
    const int by = j / perBlock, jb = j - by * perBlock;
    const int band = (k + 5 * by) & 7; // 5 = -3 mod 8: block (band, by) belongs to XCD (band + 3 by) mod 8
    const int x0 = band * cols, y0 = by * rows;
    const int wk = imin(c.tilesX - x0, cols), hk = imin(c.tilesY - y0, rows); // the last band / block row may be smaller (or empty)
    if (wk <= 0 || jb >= wk * hk)
        return false;
#ifndef NRD_STRIP_TARGET // widest column strip inside a block (tiles)
#define NRD_STRIP_TARGET 30
#endif
    const int nStrips = imax((wk + NRD_STRIP_TARGET / 2) / NRD_STRIP_TARGET, 1), S = (wk + nStrips - 1) / nStrips;
    const int perStrip = S * hk;
    const int s = jb / perStrip, r = jb - s * perStrip;
    const int w = imin(S, wk - s * S);
    const int tyl = r / w;
    tx = x0 + s * S + (r - tyl * w);
    ty = y0 + tyl + c.tileY0;
    return true;
}
// The 400 texels of a 20x20 tile (window [-2, 18)^2 around the workgroup's 16x16 pixels): the 256 interior positions ARE the threads'
// own pixels - every thread writes what its centre loads returned to window position (threadIdx + 2), no second load of the same
// texel (round 3: that sweep was 12-24 bytes per pixel through L1 for values the workgroup already held) - and threads 0..143 each
// fetch one position of the 2-texel ring: rows 0, 1, 18, 19 (80 positions), then columns 0, 1, 18, 19 of rows 2..17 (64)
NRD_DEV bool ring_pos(int tid, int& lx, int& ly) {
    if (tid >= 144)
        return false;
    if (tid < 80) { // rows 0, 1, 18, 19: all 20 columns
        const int r = (tid >= 20 ? 1 : 0) + (tid >= 40 ? 1 : 0) + (tid >= 60 ? 1 : 0);
        lx = tid - r * 20;
        ly = r + (r >= 2 ? 16 : 0);
    } else { // columns 0, 1, 18, 19 of rows 2..17
        const int k = tid - 80, q = k & 3;
        lx = q + (q >= 2 ? 16 : 0);
        ly = 2 + (k >> 2);
    }
    return true;
}

// One 8- / 16-bit word of a per-tile plane through the SCALAR data path (same address for the whole workgroup; constant address space:
// s_load_dword of the aligned word that holds it): the flag does not queue in the in-order vector memory counter with the workgroup's loads
#ifndef NRD_TILE_TEXEL // (the host emulation of the tests substitutes a plain 1- / 2-byte read - the aligned word may end up to 3 bytes behind
                       // the plane, which a device allocation always covers and an instrumented host heap does not; like NRD_WAVES_PER_EU)
#define NRD_TILE_TEXEL(addr, bytes)                                                                                           \
    ((*(const __attribute__((address_space(4))) uint32_t*)((addr) & ~(uintptr_t)3) >> ((uint32_t)((addr) & (uintptr_t)(4 - (bytes))) * 8u)) & \
     ((bytes) == 1 ? 0xffu : 0xffffu))
#endif
NRD_DEV uint32_t ld_tile_u16(const PlaneRef& P, int tx, int ty) { return NRD_TILE_TEXEL((uintptr_t)P.p + texel_offset(P, tx, ty, 2, 0), 2); }
NRD_DEV uint32_t ld_tile_u8(const PlaneRef& P, int tx, int ty) { return NRD_TILE_TEXEL((uintptr_t)P.p + texel_offset(P, tx, ty, 1, 0), 1); }

// One dword through the scalar data path (uniform address, read-only for the launch): constant address space = s_load_dword
#ifndef NRD_SCALAR_U32 // (the host emulation of the tests reads the plain word)
#define NRD_SCALAR_U32(ptr) (*(const __attribute__((address_space(4))) uint32_t*)(uintptr_t)(ptr))
#endif
// The kernel arguments once more, as values the compiler cannot tie to the copies it already holds: a second-half-of-the-kernel reads its
// constants with fresh s_loads where it needs them instead of keeping the first half's wide loads alive in between (the fused PrePass +
// TemporalAccumulation kernel held 32 SGPRs of camera matrices across its tap loop for the reprojection behind it and ran out of scalar
// registers). The host emulation of the tests defines it as the identity.
#ifndef NRD_RELOAD_ARGS
#define NRD_RELOAD_ARGS(T, p, q)                                                                 \
    const __attribute__((address_space(4))) T* q##_ptr = (const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr(); /* `p` is the kernel's only argument */ \
    asm volatile("" : "+s"(q##_ptr));                                                            \
    const T& q = *(const T*)q##_ptr /* (the address-space inference pass turns loads through it back into constant-address loads) */
#endif
#ifndef NRD_TILE_TABLE // 0: every wave computes its tile (xcd_tile_kj) - the A/B switch of profiles/r05_ab_tile_table.txt
#define NRD_TILE_TABLE 1
#endif
#ifndef NRD_TILE_FLAGS // 0: the tile flag always comes from the Tiles plane (a scalar load that depends on the table entry)
#define NRD_TILE_FLAGS 1
#endif
// the j-th tile of XCD k in this launch's direction: from the table when the launch has one. `tflag` = the tile's ClassifyTiles flag when
// the launch carries the flags in table order (FrameConsts::tileFlags), -1 when it must be read from the Tiles plane
// Values that must sit in scalar registers HERE: the kernel-argument loads behind them are issued together, in front of this point - one
// round trip - instead of one by one where the compiler first needs each (it sinks them behind the branches: a wave of Blur made four
// dependent trips to the argument segment before its first vector load). Later uses of the same arguments reuse the registers. The host
// emulation defines the macros away.
#ifndef NRD_PIN_SGPRS8
#define NRD_PIN_SGPRS8(a, b, c, d, e, f, g, h) asm volatile("" ::"s"(a), "s"(b), "s"(c), "s"(d), "s"(e), "s"(f), "s"(g), "s"(h))
#endif
#ifndef NRD_PIN_PLANES // plane pointers + pitches a kernel's first vector loads need: pinned at kernel entry, they travel with the tile look-up's arguments
#define NRD_PIN_PLANES3(A, B, C) asm volatile("" ::"s"((A).p), "s"((A).pitch), "s"((B).p), "s"((B).pitch), "s"((C).p), "s"((C).pitch))
#define NRD_PIN_PLANES4(A, B, C, D) asm volatile("" ::"s"((A).p), "s"((A).pitch), "s"((B).p), "s"((B).pitch), "s"((C).p), "s"((C).pitch), "s"((D).p), "s"((D).pitch))
#define NRD_PIN_PLANES 1
#endif
#ifndef NRD_PIN_ARGS // 0: the A/B switch of profiles/r05_ab_pinned_arguments.txt (arguments load where the compiler puts them)
#define NRD_PIN_ARGS 1
#endif
// (`pin` false: a kernel at its register limit leaves the arguments where the compiler loads them)
NRD_DEV bool xcd_tile_of(const FrameConsts& c, const int k, const int j, int& tx, int& ty, int& tflag, const bool pin = true) {
    tflag = -1;
    if (NRD_TILE_TABLE) {
        const uint32_t* const table = c.tileTable;
        const uint8_t* const flags = c.tileFlags;
        const int perXcd = c.tilesPerXcd, reverse = c.reverse, tileY0 = c.tileY0;
        if (NRD_PIN_ARGS && pin) { // (W and the owned rows: what the caller tests its pixel against right after)
            const int W = c.W, ownY0 = c.ownY0, ownY1 = c.ownY1;
            NRD_PIN_SGPRS8(table, flags, perXcd, reverse, tileY0, W, ownY0, ownY1);
        }
        if (table) {
            const int jd = reverse ? perXcd - 1 - j : j;
            const uint32_t e = NRD_SCALAR_U32(table + (jd * 8 + k));
            if (NRD_TILE_FLAGS && flags)
                tflag = (int)NRD_TILE_TEXEL((uintptr_t)flags + (uint32_t)(jd * 8 + k), 1);
            tx = (int)(e & 0xffffu);
            ty = (int)(e >> 16) + tileY0;
            return e != 0xffffffffu;
        }
    }
    return xcd_tile_kj(c, k, j, tx, ty);
}
NRD_DEV bool xcd_tile_of(const FrameConsts& c, const int k, const int j, int& tx, int& ty) {
    int tflag;
    return xcd_tile_of(c, k, j, tx, ty, tflag);
}
NRD_DEV bool xcd_tile(const FrameConsts& c, int& tx, int& ty, int& tflag, const bool pin = true) { // one 16 x 16 workgroup per tile
    const int b = (int)blockIdx.x;
    return xcd_tile_of(c, b & 7, b >> 3, tx, ty, tflag, pin);
}
NRD_DEV bool xcd_tile(const FrameConsts& c, int& tx, int& ty) {
    int tflag;
    return xcd_tile(c, tx, ty, tflag);
}

} // namespace nrdhip
