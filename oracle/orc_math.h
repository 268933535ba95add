// orc_math.h - scalar math of the CPU ORACLE (test infrastructure, NOT product code).
//
// PARITY UNPINNED vs upstream NRD: the reference tree does not contain the NRD sources
// (External/NRD is an empty submodule) and holds no golden vectors for this path,
// so this oracle is the build's frozen restatement of the algorithm; see DESIGN.md.
//
// Every function here is plain IEEE-754 binary32 arithmetic (+ - * / sqrt, compares, integer bit
// operations) evaluated in source order; the library is compiled with -ffp-contract=off so the HIP
// kernels (also contract-off) can match bit for bit. No libm transcendental is used anywhere on the
// pixel path: exp2/log2/atan are the polynomials below.
//
// Encodings restated here follow the reference call sites:
//   normal/roughness/materialID pack .. Shaders/TraceOpaque.cs.hlsl:657, Shaders/Composition.cs.hlsl:40
//   REBLUR hit distance normalisation . Shaders/TraceOpaque.cs.hlsl:421, Shaders/DlssBefore.cs.hlsl:53-54
//   radiance+hitDist pack (YCoCg) ..... Shaders/TraceOpaque.cs.hlsl:756-757, Shaders/Composition.cs.hlsl:165-166
//   GetSpecMagicCurve ................. Shaders/Shared.hlsli:305-311 ("Taken out from NRD")
//   SIGMA penumbra / translucency ..... Shaders/TraceOpaque.cs.hlsl:800-801, Shaders/Composition.cs.hlsl:60-64
#pragma once

#include <cstdint>
#include <cstring>
#include <cmath>

namespace orc {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static inline float fmin2(float a, float b) { return a < b ? a : b; }
static inline float fmax2(float a, float b) { return a > b ? a : b; }
static inline float sat(float x) { return fmin2(fmax2(x, 0.0f), 1.0f); }
static inline float clampf(float x, float a, float b) { return fmin2(fmax2(x, a), b); }
// fused multiply-add: ONE rounding (hardware FMA on both sides: x86 -mfma / v_fma_f32); everything else stays unfused
static inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---- software reciprocal / reciprocal square root (numerics contract) ---------------------------------------------
// The kernels avoid IEEE divide / sqrt in the per-pixel set-up (quarter-rate v_rcp_f32 / v_sqrt_f32 + fix-up code on gfx950);
// these plain fma sequences are the frozen definition both sides execute bit for bit: magic-constant seed + 3 Newton steps.
// rcp_ is correctly rounded for > 99.99 % of inputs (max 0.5 ulp), rsqrt_ is good to 1.3e-7 relative.
// Domain: positive, normal, finite x (the call sites guarantee it).
static inline float rcp_(float x) {
    float r = u2f(0x7EF311C7u - f2u(x));
    r = fma_(r, fma_(-x, r, 1.0f), r);
    r = fma_(r, fma_(-x, r, 1.0f), r);
    r = fma_(r, fma_(-x, r, 1.0f), r);
    return r;
}
static inline float rcps_(float x) { // any sign
    float r = rcp_(x < 0.0f ? -x : x);
    return x < 0.0f ? -r : r;
}
static inline float rsqrt_(float x) {
    float h = 0.5f * x;
    float r = u2f(0x5F3759DFu - (f2u(x) >> 1));
    r = r * fma_(-(h * r), r, 1.5f);
    r = r * fma_(-(h * r), r, 1.5f);
    r = r * fma_(-(h * r), r, 1.5f);
    return r;
}
static inline float sqrt_(float x) { return x * rsqrt_(x); } // sqrt_(0) = 0
// x^(1/3), x positive / normal / finite (ledger row 19: TAA's pow(x, 0.333333), Shaders/Taa.cs.hlsl:45-47): Newton on the inverse cube
// root, y <- y (4 - x y^3) / 3, three steps from a magic-constant seed, then x y^2
static inline float cbrt_pos_(float x) {
    float y = u2f(0x54A21D2Au - f2u(x) / 3u);
    const float c = x * (1.0f / 3.0f);
    y = y * fma_(-c, (y * y) * y, 4.0f / 3.0f);
    y = y * fma_(-c, (y * y) * y, 4.0f / 3.0f);
    y = y * fma_(-c, (y * y) * y, 4.0f / 3.0f);
    return (x * y) * y;
}
// NRD_HW_TRANSCENDENTALS = 1 (liboracle_hwt.so, the checker of libnrdhip_hwt.so; csrc/nrd_device.h has the rule): the WEIGHT-CLASS
// reciprocals / square roots / exponentials of the spatial filters are the GPU's transcendental instructions there (1 ULP each); here
// they are the correctly rounded IEEE results at the same places - the two sides then differ by roundings of weights only, and the
// flavour's parity bar is <= 1 ULP fp16 instead of bit identity. Everything a discrete decision hangs on keeps the exact sequences.
#ifndef NRD_HW_TRANSCENDENTALS
#define NRD_HW_TRANSCENDENTALS 0
#endif
static const bool HW_TRANSCENDENTALS = NRD_HW_TRANSCENDENTALS != 0;
static inline float wrcp_(float x) { return HW_TRANSCENDENTALS ? 1.0f / x : rcp_(x); }
static inline float wsqrt_(float x) { return HW_TRANSCENDENTALS ? sqrtf(x) : sqrt_(x); }
static inline float lerpf(float a, float b, float t) { return fma_(b - a, t, a); }
static inline float smoothstep01(float x) { x = sat(x); return x * x * (3.0f - 2.0f * x); }
static inline float absf(float x) { return __builtin_fabsf(x); }

static inline f3 add3(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline f3 mul3(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float dot3(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
static inline f3 cross3(f3 a, f3 b) { return {fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))}; }
static inline f3 normalize3(f3 a) {
    float l2 = dot3(a, a);
    float inv = rsqrt_(fmax2(l2, 1e-30f));
    return mul3(a, inv);
}
// 3x3 matrix (row-major m[r*3+c]) times vector
static inline f3 rot3(const float* m, f3 v) {
    return {fma_(m[2], v.z, fma_(m[1], v.y, m[0] * v.x)), fma_(m[5], v.z, fma_(m[4], v.y, m[3] * v.x)), fma_(m[8], v.z, fma_(m[7], v.y, m[6] * v.x))};
}

// ---------------------------------------------------------------------------------------------
// fp16 <-> fp32, round-to-nearest-even, denormals honoured (matches v_cvt_f16_f32 / v_cvt_f32_f16)
// ---------------------------------------------------------------------------------------------
static inline uint16_t f32_to_f16(float f) {
    uint32_t u = f2u(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) // inf / nan
        return (uint16_t)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));
    if (u >= 0x477ff000u) // >= 65520 rounds to inf
        return (uint16_t)(sign | 0x7c00u);
    if (u < 0x38800000u) { // below the smallest normal half (2^-14): denormal or zero
        if (u < 0x33000000u) // < 2^-25 -> 0 (2^-25 itself ties to even = 0)
            return (uint16_t)sign;
        uint32_t e = u >> 23;
        uint32_t m = (u & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126u - e; // 14..24
        uint32_t h = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (h & 1u)))
            h++;
        return (uint16_t)(sign | h);
    }
    uint32_t v = u - 0x38000000u; // rebias exponent 127 -> 15
    uint32_t h = v >> 13;
    uint32_t rem = v & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u)))
        h++;
    return (uint16_t)(sign | h);
}

static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0)
            return u2f(sign);
        // denormal: m * 2^-24
        float v = (float)m * 5.9604644775390625e-08f;
        return u2f(f2u(v) | sign);
    }
    if (e == 31)
        return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

static const float FP16_MAX = 65504.0f;

// ---------------------------------------------------------------------------------------------
// Polynomial transcendentals (frozen; the HIP kernels carry the same coefficients)
// ---------------------------------------------------------------------------------------------
// 2^x, x clamped to [-126, 126]; reduction to [-0.5, 0.5], degree-6 polynomial (rel. err ~1e-7)
static inline float exp2_poly(float x) {
    x = clampf(x, -126.0f, 126.0f);
    float fi = floorf(x + 0.5f);
    float f = x - fi;
    float p = 1.535336188319500e-4f;
    p = fma_(p, f, 1.339887440266574e-3f);
    p = fma_(p, f, 9.618437357674640e-3f);
    p = fma_(p, f, 5.550332471162809e-2f);
    p = fma_(p, f, 2.402264791363012e-1f);
    p = fma_(p, f, 6.931472028550421e-1f);
    p = fma_(p, f, 1.0f);
    int32_t e = (int32_t)fi;
    float scale = u2f((uint32_t)(e + 127) << 23);
    return p * scale;
}

// log2(x) for normal positive x (x <= 0 returns -126)
static inline float log2_poly(float x) {
    if (!(x > 1.17549435e-38f))
        return -126.0f;
    uint32_t u = f2u(x);
    int32_t e = (int32_t)((u >> 23) & 0xffu) - 127;
    float m = u2f((u & 0x7fffffu) | 0x3f800000u); // [1, 2)
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

// pow(saturate(x), y), y >= 0
static inline float pow01(float x, float y) {
    x = sat(x);
    if (x <= 0.0f)
        return 0.0f;
    return exp2_poly(y * log2_poly(x));
}

// atan(x), x >= 0 (Abramowitz-Stegun 4.4.49 on [0,1], reflected above 1)
static inline float atan_pos(float x) {
    bool inv = x > 1.0f;
    float t = inv ? wrcp_(x) : x;
    float s = t * t;
    float p = 0.0208351f;
    p = fma_(p, s, -0.0851330f);
    p = fma_(p, s, 0.1801410f);
    p = fma_(p, s, -0.3302995f);
    p = fma_(p, s, 0.9998660f);
    p = p * t;
    return inv ? 1.57079633f - p : p;
}

// acos(x) ~ sqrt(2) * sqrt(1 - x), x in [0, 1] (small-angle exact, monotonic)
static inline float acos_approx(float x) { return 1.41421356f * sqrtf(sat(1.0f - x)); }

// NRD_UPSTREAM_FORMULAS = 1: the build flavour with the RECALLED upstream forms of ledger rows 1, 2, 7 and 13 (oracle/README.md):
// hit-distance weight exp(-3 |x|), normal weight on the angle, Blur rotation per pixel, RELAX in linear RGB - liboracle.so (the default since round 4), the checker of
// libnrdhip.so (same switch, same formulas: nrd-sample_amd/csrc/nrd_device.h)
#ifndef NRD_UPSTREAM_FORMULAS // (1 = the default since round 4; 0 = the "frozen" flavour: liboracle_frozen.so, the checker of libnrdhip_frozen.so)
#define NRD_UPSTREAM_FORMULAS 1
#endif
static const bool UPSTREAM_FORMULAS = NRD_UPSTREAM_FORMULAS != 0;
static const int BLUR_ROTATION_SHIFT = NRD_UPSTREAM_FORMULAS ? 0 : 1; // Blur's Poisson rotation: per pixel (upstream) / per 2x2 quad (frozen)

// The angle between two normals as upstream's weights take it: Math::AcosApprox(cos) = sqrt(2) sqrt(saturate(1 - cos)) (MathLib, recalled)
// = the CHORD of the two unit vectors = (2 / 1023) sqrt(d2) on guide normals, d2 = squared distance of the 10-bit codes (orc_core.h
// normal_dist2). The square root: magic seed + ONE Newton step with tuned constants, relative error <= 6.5e-4 - as fine as the quantised
// d2 deserves (csrc/nrd_device.h sqrt1_unscaled_); returns sqrt(x) / SQRT1_SCALE
static const float SQRT1_SCALE = 0.703952253f;
static inline float sqrt1_unscaled_(float x) {
    const float y = u2f(0x5F1FFFF9u - (f2u(x) >> 1));
    const float u = x * y;
    return u * fma_(-u, y, 2.38924456f);
}
static const float NORMAL_CHORD_SCALE = (2.0f / 1023.0f) * (HW_TRANSCENDENTALS ? 1.0f : SQRT1_SCALE);
static inline float chord_unscaled_(float d2) { return HW_TRANSCENDENTALS ? sqrtf(d2) : sqrt1_unscaled_(d2); }
// 2^x for x <= 0, degree-3 polynomial after a round-to-nearest split (relative error 8.0e-5): csrc/nrd_device.h exp2_poly_neg
static inline float exp2_poly_neg(float x) {
    x = fmax2(x, -126.0f);
    const float fi = rintf(x); // round to nearest even (the default rounding mode; the kernels: v_rndne_f32)
    const float f = x - fi;
    float p = 5.519811809062958e-2f;
    p = fma_(p, f, 2.4267692863941193e-1f);
    p = fma_(p, f, 6.932618021965027e-1f);
    p = fma_(p, f, 9.999227523803711e-1f);
    return ldexpf(p, (int)fi);
}
static inline float exp2_neg(float x) { return HW_TRANSCENDENTALS ? exp2f(x) : exp2_poly_neg(x); }
// hit-distance weight: compact-support stand-in for exp(-3 |x|): (1 - |x|)^2 clamped (division-free); upstream flavour: exp(-3 |x|)
static const float EXP_WEIGHT_SCALE = UPSTREAM_FORMULAS ? 4.32808512f : 1.0f; // 3 log2(e)
static inline float exp_weight(float ax) {
    if (UPSTREAM_FORMULAS)
        return exp2_neg(-EXP_WEIGHT_SCALE * ax);
    float t = sat(1.0f - ax);
    return t * t;
}
// the same weight of |v| with EXP_WEIGHT_SCALE already folded into v (the spatial passes' taps)
static inline float exp_weight_prescaled(float v) {
    if (UPSTREAM_FORMULAS)
        return exp2_neg(-absf(v));
    float t = sat(1.0f - absf(v));
    return t * t;
}
// normal weight from the squared distance d2 of two normals' 10-bit codes (orc_core.h normal_dist2; 1 - cos = d2 NORMAL_D2_TO_1MCOS).
// Frozen: on the SQUARED angle, angle^2 ~ 2 (1 - cos) (sqrt-free), parameter w2 = 1 / angleMax^2; default flavour: smoothstep(1 - angle /
// angleMax) on the chord (sqrt1_unscaled_ above), parameter 1 / angleMax
static const float NORMAL_D2_TO_1MCOS = 0.5f * (2.0f / 1023.0f) * (2.0f / 1023.0f);
static inline float nw_param(float normalW) { return UPSTREAM_FORMULAS ? normalW : normalW * normalW; }
static inline float normal_weight(float d2, float prm) {
    if (UPSTREAM_FORMULAS)
        return smoothstep01(fma_(-chord_unscaled_(d2), prm * NORMAL_CHORD_SCALE, 1.0f));
    return smoothstep01(fma_(-2.0f * sat(1.0f - fma_(d2, -NORMAL_D2_TO_1MCOS, 1.0f)), prm, 1.0f));
}

// ---------------------------------------------------------------------------------------------
// Packing
// ---------------------------------------------------------------------------------------------
static inline float sign_nz(float v) { return v >= 0.0f ? 1.0f : -1.0f; }

// octahedral unit vector <-> [0,1]^2
static inline f2 oct_encode(f3 n) {
    float inv = 1.0f / (absf(n.x) + absf(n.y) + absf(n.z));
    float x = n.x * inv, y = n.y * inv;
    if (n.z < 0.0f) {
        float ox = (1.0f - absf(y)) * sign_nz(x);
        float oy = (1.0f - absf(x)) * sign_nz(y);
        x = ox;
        y = oy;
    }
    return {x * 0.5f + 0.5f, y * 0.5f + 0.5f};
}

static inline f3 oct_decode(f2 p) {
    float fx = p.x * 2.0f - 1.0f, fy = p.y * 2.0f - 1.0f;
    float nz = 1.0f - absf(fx) - absf(fy);
    float t = sat(-nz);
    float nx = fx + (fx >= 0.0f ? -t : t);
    float ny = fy + (fy >= 0.0f ? -t : t);
    return normalize3({nx, ny, nz});
}

static inline uint32_t unorm_bits(float v, float maxv) { return (uint32_t)floorf(sat(v) * maxv + 0.5f); }

// NRD_FrontEnd_PackNormalAndRoughness, NRD_NORMAL_ENCODING = 2 (R10G10B10A2), roughness LINEAR
static inline uint32_t pack_normal_roughness(f3 n, float roughness, uint32_t materialID) {
    f2 o = oct_encode(n);
    uint32_t x = unorm_bits(o.x, 1023.0f), y = unorm_bits(o.y, 1023.0f), z = unorm_bits(roughness, 1023.0f);
    return x | (y << 10) | (z << 20) | ((materialID & 3u) << 30);
}

struct NormalRoughness {
    f3 n;
    float roughness;
    uint32_t materialID;
};

static inline NormalRoughness unpack_normal_roughness(uint32_t p) {
    NormalRoughness r;
    f2 o = {(float)(p & 1023u) / 1023.0f, (float)((p >> 10) & 1023u) / 1023.0f};
    r.n = oct_decode(o);
    r.roughness = (float)((p >> 20) & 1023u) / 1023.0f;
    r.materialID = p >> 30;
    return r;
}

static inline f3 linear_to_ycocg(f3 c) {
    float Y = c.x * 0.25f + c.y * 0.5f + c.z * 0.25f;
    float Co = c.x * 0.5f - c.z * 0.5f;
    float Cg = c.y * 0.5f - c.x * 0.25f - c.z * 0.25f;
    return {Y, Co, Cg};
}

static inline f3 ycocg_to_linear(f3 c) {
    float t = c.x - c.z;
    f3 r = {t + c.y, c.x + c.z, t - c.y};
    return {fmax2(r.x, 0.0f), fmax2(r.y, 0.0f), fmax2(r.z, 0.0f)};
}

// ledger row 13 (NRD_UPSTREAM_FORMULAS flavour): RELAX keeps linear RGB internally; the luminance of a RELAX texel is then Rec.709 of
// its rgb instead of channel 0 (Y of YCoCg), and a luminance clamp scales all three channels
static const bool RELAX_LINEAR_RGB = UPSTREAM_FORMULAS;
static inline float luma709(f4 v) { return fma_(v.x, 0.2126f, fma_(v.y, 0.7152f, v.z * 0.0722f)); }
static inline float signal_luma(f4 v, bool relax) { return (RELAX_LINEAR_RGB && relax) ? luma709(v) : v.x; }
static inline void clamp_luma(f4& v, float Yc, float scale, bool relax) {
    v.x = (RELAX_LINEAR_RGB && relax) ? v.x * scale : Yc;
    v.y *= scale;
    v.z *= scale;
}

// (1 - 2^(-200 r^2)) * sqrt(r)   [Shaders/Shared.hlsli:305-311]
static inline float spec_magic_curve(float roughness) {
    float f = 1.0f - exp2_poly(-200.0f * roughness * roughness);
    return f * sqrt_(sat(roughness));
}

// REBLUR hit distance normalisation: (A + |z| B) * lerp(1, C, 2^(D r^2))
static inline float reblur_hitdist_factor(const float* hp, float roughness) { return lerpf(1.0f, hp[2], exp2_poly(hp[3] * roughness * roughness)); } // the roughness-dependent factor
static inline float reblur_hitdist_norm(float absViewZ, const float* hp, float roughness) { return fma_(absViewZ, hp[1], hp[0]) * reblur_hitdist_factor(hp, roughness); }

// specular lobe half angle: atan(r^2 * k / (1 - k)), k = 0.75
static inline float spec_lobe_half_angle(float roughness) {
    float m = sat(roughness);
    m = m * m;
    return atan_pos(m * 3.0f);
}

// Frostbite-style dominant direction factor
static inline float spec_dominant_factor(float roughness) {
    float s = sat(1.0f - roughness);
    return s * (sqrt_(s) + roughness);
}

// integer hash -> rotation table index
static inline uint32_t hash_px(uint32_t x, uint32_t y, uint32_t frame, uint32_t salt) {
    uint32_t h = (x * 73856093u) ^ (y * 19349663u) ^ (frame * 83492791u) ^ (salt * 2654435761u);
    h ^= h >> 13;
    h *= 0x5bd1e995u;
    h ^= h >> 15;
    return h;
}

// tangent basis of a unit vector (branchless Frisvad/Duff form)
static inline void basis3(f3 n, f3& t, f3& b) {
    float sz = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = -rcps_(sz + n.z);
    float nxa = n.x * a, sx = sz * n.x;
    float bb = nxa * n.y;
    t = {fma_(sx, nxa, 1.0f), sz * bb, -sx};
    b = {bb, fma_(n.y * a, n.y, sz), -n.y};
}

} // namespace orc
