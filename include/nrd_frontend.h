// nrd_frontend.h - the shader-side helper API of NRD (upstream: Shaders/Include/NRD.hlsli of the absent External/NRD submodule)
// for producers and consumers written in HIP or plain C++: what the sample's path tracer calls to ENCODE the denoiser inputs
// and what its composition pass calls to DECODE the outputs. Host + device: every function is `__host__ __device__` under
// hipcc and a plain inline function under a host compiler, so a CPU producer packs bit-identically to a GPU one.
//
// Call sites in the reference (paths relative to /root/reference):
//   NRD_FrontEnd_PackNormalAndRoughness ............ Shaders/TraceOpaque.cs.hlsl:657
//   NRD_FrontEnd_UnpackNormalAndRoughness ........... Shaders/Composition.cs.hlsl:93-96
//   REBLUR_FrontEnd_GetNormHitDist .................. Shaders/TraceOpaque.cs.hlsl:421
//   NRD_FrontEnd_SpecHitDistAveraging_{Begin,Add,End} Shaders/TraceOpaque.cs.hlsl:99, :437, :465
//   REBLUR_/RELAX_FrontEnd_PackRadianceAnd*HitDist .. Shaders/TraceOpaque.cs.hlsl:743-744, :756-757
//   REBLUR_/RELAX_FrontEnd_PackSh ................... Shaders/TraceOpaque.cs.hlsl:740-741, :751-752
//   REBLUR_FrontEnd_PackDirectionalOcclusion ........ Shaders/TraceOpaque.cs.hlsl:754
//   SIGMA_FrontEnd_PackPenumbra / PackTranslucency .. Shaders/TraceOpaque.cs.hlsl:800-801
//   NRD_MaterialFactors ............................. Shaders/RaytracingShared.hlsli:929, :946 (de-modulation), Composition.cs.hlsl:183-188
//   REBLUR_/RELAX_BackEnd_Unpack*, SIGMA_BackEnd_UnpackShadow, NRD_SG_* ... Shaders/Composition.cs.hlsl:57-64, :74-175
//
// PARITY: the bodies are this build's frozen definitions (the upstream header is not in the reference tree); encodings match
// what the denoiser kernels of this library read (include/NRDSettings.h NRD_NORMAL_ENCODING = R10G10B10A2, linear roughness;
// YCoCg radiance for REBLUR, linear RGB for RELAX; SH1 = direction x luma). oracle/README.md lists where a definition is a
// recollection of upstream rather than a citation.
#ifndef NRD_FRONTEND_H
#define NRD_FRONTEND_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NRD_FE __host__ __device__ static inline
#else
#define NRD_FE static inline
#endif

namespace nrd_fe {

struct float3_ {
    float x, y, z;
};
struct float4_ {
    float x, y, z, w;
};

#define NRD_FP16_MAX 65504.0f

NRD_FE float fe_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
NRD_FE float fe_min(float a, float b) { return a < b ? a : b; }
NRD_FE float fe_max(float a, float b) { return a > b ? a : b; }
NRD_FE float fe_sat(float x) { return fe_min(fe_max(x, 0.0f), 1.0f); }
NRD_FE float fe_abs(float x) { return x < 0.0f ? -x : x; }
NRD_FE float fe_dot(float3_ a, float3_ b) { return fe_fma(a.z, b.z, fe_fma(a.y, b.y, a.x * b.x)); }
NRD_FE uint32_t fe_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
NRD_FE float fe_float(uint32_t u) { return __builtin_bit_cast(float, u); }
NRD_FE bool fe_finite(float x) { return (fe_bits(x) & 0x7f800000u) != 0x7f800000u; }
NRD_FE float3_ fe_normalize(float3_ v) {
    float l = __builtin_sqrtf(fe_max(fe_dot(v, v), 1e-30f));
    return {v.x / l, v.y / l, v.z / l};
}

// 2^x, fixed degree-6 polynomial on [-1/2, 1/2] (the coefficients the kernels use, csrc/nrd_device.h exp2_poly): identical bits on
// host and device, no libm
NRD_FE float fe_exp2(float x) {
    x = fe_min(fe_max(x, -126.0f), 126.0f);
    float fi = __builtin_floorf(x + 0.5f);
    float f = x - fi;
    float p = 1.535336188319500e-4f;
    p = fe_fma(p, f, 1.339887440266574e-3f);
    p = fe_fma(p, f, 9.618437357674640e-3f);
    p = fe_fma(p, f, 5.550332471162809e-2f);
    p = fe_fma(p, f, 2.402264791363012e-1f);
    p = fe_fma(p, f, 6.931472028550421e-1f);
    p = fe_fma(p, f, 1.0f);
    return p * fe_float((uint32_t)((int)fi + 127) << 23);
}
// log2(x) for x > 0 (Cephes logf polynomial, same as csrc/nrd_device.h log2_poly)
NRD_FE float fe_log2(float x) {
    if (!(x > 1.17549435e-38f))
        return -126.0f;
    uint32_t u = fe_bits(x);
    int e = (int)((u >> 23) & 0xffu) - 127;
    float m = fe_float((u & 0x7fffffu) | 0x3f800000u);
    if (m > 1.41421356f) {
        m = m * 0.5f;
        e += 1;
    }
    float t = m - 1.0f, z = t * t;
    float p = 7.0376836292e-2f;
    p = fe_fma(p, t, -1.1514610310e-1f);
    p = fe_fma(p, t, 1.1676998740e-1f);
    p = fe_fma(p, t, -1.2420140846e-1f);
    p = fe_fma(p, t, 1.4249322787e-1f);
    p = fe_fma(p, t, -1.6668057665e-1f);
    p = fe_fma(p, t, 2.0000714765e-1f);
    p = fe_fma(p, t, -2.4999993993e-1f);
    p = fe_fma(p, t, 3.3333331174e-1f);
    float y = t * z * p;
    y = y - 0.5f * z;
    return (t + y) * 1.44269504f + (float)e;
}
NRD_FE float fe_pow(float x, float y) { return x <= 0.0f ? 0.0f : fe_exp2(y * fe_log2(x)); }

// fp16 bit pattern of a float, round to nearest even, clamped to +-NRD_FP16_MAX (what a RGBA16_SFLOAT store does to the value)
NRD_FE uint16_t NRD_FloatToHalf(float f) {
    f = fe_min(fe_max(f, -NRD_FP16_MAX), NRD_FP16_MAX);
    uint32_t u = fe_bits(f), sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u > 0x7f800000u)
        return (uint16_t)(sign | 0x7e00u); // NaN
    if (u < 0x33000001u)
        return (uint16_t)sign; // rounds to zero
    int e = (int)(u >> 23) - 127;
    uint32_t m = (u & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? (-14 - e) + 13 : 13; // denormal halves lose more mantissa bits
    uint32_t he = e < -14 ? 0u : (uint32_t)(e + 15);
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u)))
        q++;
    uint32_t h = e < -14 ? q : ((he << 10) + (q - 0x400u)); // a mantissa carry flows into the exponent
    return (uint16_t)(sign | h);
}
NRD_FE float NRD_HalfToFloat(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    if (e == 0) {
        if (m == 0)
            return fe_float(sign);
        float v = (float)m * 5.9604644775390625e-8f; // 2^-24
        return sign ? -v : v;
    }
    if (e == 31)
        return fe_float(sign | 0x7f800000u | (m << 13));
    return fe_float(sign | ((e + 112u) << 23) | (m << 13));
}
struct half4_ {
    uint32_t lo, hi; // RGBA16_SFLOAT texel: {x | y << 16, z | w << 16}
};
NRD_FE half4_ NRD_PackHalf4(float4_ v) {
    return {(uint32_t)NRD_FloatToHalf(v.x) | ((uint32_t)NRD_FloatToHalf(v.y) << 16), (uint32_t)NRD_FloatToHalf(v.z) | ((uint32_t)NRD_FloatToHalf(v.w) << 16)};
}
NRD_FE float4_ NRD_UnpackHalf4(half4_ t) {
    return {NRD_HalfToFloat((uint16_t)(t.lo & 0xffffu)), NRD_HalfToFloat((uint16_t)(t.lo >> 16)), NRD_HalfToFloat((uint16_t)(t.hi & 0xffffu)), NRD_HalfToFloat((uint16_t)(t.hi >> 16))};
}

// ---- G-buffer -----------------------------------------------------------------------------------------------------------
// octahedral mapping of a unit vector to [0, 1]^2 and back
NRD_FE void NRD_OctEncode(float3_ n, float& ox, float& oy) {
    float l1 = fe_abs(n.x) + fe_abs(n.y) + fe_abs(n.z);
    float x = n.x / l1, y = n.y / l1, z = n.z / l1;
    float fx = z < 0.0f ? (1.0f - fe_abs(y)) * (x >= 0.0f ? 1.0f : -1.0f) : x;
    float fy = z < 0.0f ? (1.0f - fe_abs(x)) * (y >= 0.0f ? 1.0f : -1.0f) : y;
    ox = fe_fma(fx, 0.5f, 0.5f);
    oy = fe_fma(fy, 0.5f, 0.5f);
}
NRD_FE float3_ NRD_OctDecode(float ox, float oy) {
    float fx = ox * 2.0f - 1.0f, fy = oy * 2.0f - 1.0f;
    float nz = 1.0f - fe_abs(fx) - fe_abs(fy);
    float t = fe_sat(-nz);
    float nx = fx + (fx >= 0.0f ? -t : t), ny = fy + (fy >= 0.0f ? -t : t);
    return fe_normalize({nx, ny, nz});
}
NRD_FE uint32_t fe_unorm(float v, float scale) { return (uint32_t)__builtin_floorf(fe_fma(fe_sat(v), scale, 0.5f)); }

// IN_NORMAL_ROUGHNESS texel (R10_G10_B10_A2_UNORM): octahedral world normal, linear roughness, materialID in [0, 3]
NRD_FE uint32_t NRD_FrontEnd_PackNormalAndRoughness(float3_ N, float roughness, float materialID = 0.0f) {
    float ox, oy;
    NRD_OctEncode(N, ox, oy);
    uint32_t m = (uint32_t)fe_min(fe_max(materialID, 0.0f), 3.0f);
    return fe_unorm(ox, 1023.0f) | (fe_unorm(oy, 1023.0f) << 10) | (fe_unorm(roughness, 1023.0f) << 20) | (m << 30);
}
// -> {N.xyz, roughness}; materialID through the reference parameter
NRD_FE float4_ NRD_FrontEnd_UnpackNormalAndRoughness(uint32_t p, float& materialID) {
    float3_ n = NRD_OctDecode((float)(p & 1023u) / 1023.0f, (float)((p >> 10) & 1023u) / 1023.0f);
    materialID = (float)(p >> 30);
    return {n.x, n.y, n.z, (float)((p >> 20) & 1023u) / 1023.0f};
}
NRD_FE float4_ NRD_FrontEnd_UnpackNormalAndRoughness(uint32_t p) {
    float m;
    return NRD_FrontEnd_UnpackNormalAndRoughness(p, m);
}

// ---- radiance ------------------------------------------------------------------------------------------------------------
NRD_FE float3_ NRD_LinearToYCoCg(float3_ c) { return {c.x * 0.25f + c.y * 0.5f + c.z * 0.25f, c.x * 0.5f - c.z * 0.5f, c.y * 0.5f - c.x * 0.25f - c.z * 0.25f}; }
NRD_FE float3_ NRD_YCoCgToLinear(float3_ c) {
    float t = c.x - c.z;
    return {fe_max(t + c.y, 0.0f), fe_max(c.x + c.z, 0.0f), fe_max(t - c.y, 0.0f)};
}
NRD_FE bool NRD_IsValidRadiance(float3_ c) { return fe_finite(c.x) && fe_finite(c.y) && fe_finite(c.z); }
NRD_FE float fe_luma(float3_ c) { return c.x * 0.25f + c.y * 0.5f + c.z * 0.25f; } // the Y of YCoCg

// hit distance normalisation of REBLUR: f(viewZ, roughness) = (A + |viewZ| B) lerp(1, C, 2^(D roughness^2)); hitDistParams =
// ReblurHitDistanceParameters {A, B, C, D} (include/NRDSettings.h; the sample fills A, B, C and keeps D = -25)
NRD_FE float REBLUR_GetHitDistanceNormalization(float viewZ, const float hitDistParams[4], float roughness = 1.0f) {
    float e = fe_exp2(hitDistParams[3] * roughness * roughness);
    return fe_fma(fe_abs(viewZ), hitDistParams[1], hitDistParams[0]) * fe_fma(hitDistParams[2] - 1.0f, e, 1.0f);
}
NRD_FE float REBLUR_FrontEnd_GetNormHitDist(float hitDist, float viewZ, const float hitDistParams[4], float roughness = 1.0f) {
    return fe_sat(hitDist / REBLUR_GetHitDistanceNormalization(viewZ, hitDistParams, roughness));
}

// Specular hit distances of several paths per pixel are merged with a soft minimum, not a mean (a mean of a near and a far hit
// describes neither reflection): sum 2^(-k h) over the paths, then -log2(sum) / k. h = normalised hit distance.
#define NRD_SPEC_HITDIST_AVERAGING_K 17.0f
NRD_FE float NRD_FrontEnd_SpecHitDistAveraging_Begin() { return 0.0f; }
NRD_FE void NRD_FrontEnd_SpecHitDistAveraging_Add(float& accumulatedSpecHitDist, float hitDist) { accumulatedSpecHitDist += fe_exp2(-NRD_SPEC_HITDIST_AVERAGING_K * hitDist); }
NRD_FE void NRD_FrontEnd_SpecHitDistAveraging_End(float& accumulatedSpecHitDist) {
    accumulatedSpecHitDist = accumulatedSpecHitDist > 0.0f ? fe_max(-fe_log2(accumulatedSpecHitDist) / NRD_SPEC_HITDIST_AVERAGING_K, 0.0f) : 0.0f; // no specular path: 0
}

// IN_*_RADIANCE_HITDIST texels (store with NRD_PackHalf4). sanitize: non-finite radiance becomes 0 (USE_SANITIZATION)
NRD_FE float4_ REBLUR_FrontEnd_PackRadianceAndNormHitDist(float3_ radiance, float normHitDist, bool sanitize = true) {
    if (sanitize && !NRD_IsValidRadiance(radiance))
        radiance = {0.0f, 0.0f, 0.0f};
    if (sanitize && !fe_finite(normHitDist))
        normHitDist = 0.0f;
    float3_ c = NRD_LinearToYCoCg(radiance);
    return {fe_min(c.x, NRD_FP16_MAX), c.y, c.z, fe_sat(normHitDist)};
}
NRD_FE float4_ RELAX_FrontEnd_PackRadianceAndHitDist(float3_ radiance, float hitDist, bool sanitize = true) {
    if (sanitize && !NRD_IsValidRadiance(radiance))
        radiance = {0.0f, 0.0f, 0.0f};
    if (sanitize && !fe_finite(hitDist))
        hitDist = 0.0f;
    return {fe_min(fe_max(radiance.x, 0.0f), NRD_FP16_MAX), fe_min(fe_max(radiance.y, 0.0f), NRD_FP16_MAX), fe_min(fe_max(radiance.z, 0.0f), NRD_FP16_MAX),
            fe_min(fe_max(hitDist, 0.0f), NRD_FP16_MAX)};
}
// SH mode: SH0 = the plain texel, SH1 = {direction x luma, 0} (direction = unit vector TOWARD the light sample)
NRD_FE float4_ REBLUR_FrontEnd_PackSh(float3_ radiance, float normHitDist, float3_ direction, float4_& sh1, bool sanitize = true) {
    float4_ sh0 = REBLUR_FrontEnd_PackRadianceAndNormHitDist(radiance, normHitDist, sanitize);
    sh1 = {direction.x * sh0.x, direction.y * sh0.x, direction.z * sh0.x, 0.0f};
    return sh0;
}
NRD_FE float4_ RELAX_FrontEnd_PackSh(float3_ radiance, float hitDist, float3_ direction, float4_& sh1, bool sanitize = true) {
    float4_ sh0 = RELAX_FrontEnd_PackRadianceAndHitDist(radiance, hitDist, sanitize);
    float Y = fe_luma({sh0.x, sh0.y, sh0.z});
    sh1 = {direction.x * Y, direction.y * Y, direction.z * Y, 0.0f};
    return sh0;
}
NRD_FE float4_ REBLUR_FrontEnd_PackDirectionalOcclusion(float3_ direction, float normHitDist, bool sanitize = true) {
    if (sanitize && !fe_finite(normHitDist))
        normHitDist = 0.0f;
    float h = fe_sat(normHitDist);
    return {direction.x * h, direction.y * h, direction.z * h, h};
}

// ---- SIGMA ---------------------------------------------------------------------------------------------------------------
// distanceToOccluder: NRD_FP16_MAX (or anything >= it) = the shadow ray missed
NRD_FE float SIGMA_FrontEnd_PackPenumbra(float distanceToOccluder, float tanOfLightAngularRadius) {
    if (!(distanceToOccluder < NRD_FP16_MAX))
        return NRD_FP16_MAX;
    return fe_min(distanceToOccluder * tanOfLightAngularRadius, 32768.0f);
}
NRD_FE float4_ SIGMA_FrontEnd_PackTranslucency(float distanceToOccluder, float3_ translucency) {
    return {distanceToOccluder < NRD_FP16_MAX ? 0.0f : 1.0f, fe_sat(translucency.x), fe_sat(translucency.y), fe_sat(translucency.z)};
}
NRD_FE uint32_t NRD_PackUnorm8x4(float4_ v) { return fe_unorm(v.x, 255.0f) | (fe_unorm(v.y, 255.0f) << 8) | (fe_unorm(v.z, 255.0f) << 16) | (fe_unorm(v.w, 255.0f) << 24); }
// OUT_SHADOW_TRANSLUCENCY stores sqrt(visibility): {shadow, translucency.rgb}
NRD_FE float4_ SIGMA_BackEnd_UnpackShadow(float4_ s) { return {s.x * s.x, s.y * s.y, s.z * s.z, s.w * s.w}; }

// ---- back end ------------------------------------------------------------------------------------------------------------
NRD_FE float4_ REBLUR_BackEnd_UnpackRadianceAndNormHitDist(float4_ v) {
    float3_ c = NRD_YCoCgToLinear({v.x, v.y, v.z});
    return {c.x, c.y, c.z, v.w};
}
NRD_FE float4_ RELAX_BackEnd_UnpackRadiance(float4_ v) { return v; }

// material (de)modulation: the denoisers work on radiance divided by these factors, the composition multiplies them back.
// Environment BRDF term: polynomial fit of Ray Tracing Gems ch. 32 ("Accurate real-time specular reflections with radiance caching")
NRD_FE float3_ NRD_EnvironmentTerm_Rtg(float3_ Rf0, float NoV, float linearRoughness) {
    float m = linearRoughness * linearRoughness;
    float X1 = NoV, X2 = NoV * NoV, X3 = NoV * X2;
    float Y1 = m, Y2 = m * m, Y3 = m * Y2;
    // bias = (M1 [1, NoV]) . [1, m] / ((M2 [1, NoV, NoV^3]) . [1, m, m^3]); scale likewise with M3 / M4 over [1, NoV^2, NoV^3]
    float b0 = fe_fma(-1.28514f, X1, 0.99044f), b1 = fe_fma(-0.755907f, X1, 1.29678f);
    float bn = fe_fma(b1, Y1, b0);
    float d0 = fe_fma(59.4188f, X3, fe_fma(2.92338f, X1, 1.0f)), d1 = fe_fma(222.592f, X3, fe_fma(-27.0302f, X1, 20.3225f)), d2 = fe_fma(316.627f, X3, fe_fma(626.13f, X1, 121.563f));
    float bd = fe_fma(d2, Y3, fe_fma(d1, Y1, d0));
    float s0 = fe_fma(3.32707f, X1, 0.0365463f), s1 = fe_fma(-9.04756f, X1, 9.0632f);
    float sn = fe_fma(s1, Y1, s0);
    float e0 = fe_fma(-1.36772f, X3, fe_fma(3.59685f, X2, 1.0f)), e1 = fe_fma(9.22949f, X3, fe_fma(-16.3174f, X2, 9.04401f)), e2 = fe_fma(-20.2123f, X3, fe_fma(19.7886f, X2, 5.56589f));
    float sd = fe_fma(e2, Y3, fe_fma(e1, Y1, e0));
    float bias = bn / bd, scale = sn / sd;
    return {fe_sat(fe_fma(Rf0.x, scale, bias)), fe_sat(fe_fma(Rf0.y, scale, bias)), fe_sat(fe_fma(Rf0.z, scale, bias))};
}
NRD_FE void NRD_MaterialFactors(float3_ N, float3_ V, float3_ albedo, float3_ Rf0, float roughness, float3_& diffFactor, float3_& specFactor) {
    float NoV = fe_abs(fe_dot(N, V));
    float3_ Fenv = NRD_EnvironmentTerm_Rtg(Rf0, NoV, roughness);
    diffFactor = {fe_fma((1.0f - Fenv.x) * albedo.x, 0.99f, 0.01f), fe_fma((1.0f - Fenv.y) * albedo.y, 0.99f, 0.01f), fe_fma((1.0f - Fenv.z) * albedo.z, 0.99f, 0.01f)};
    specFactor = {fe_fma(Fenv.x, 0.99f, 0.01f), fe_fma(Fenv.y, 0.99f, 0.01f), fe_fma(Fenv.z, 0.99f, 0.01f)};
}
// BRDF::ConvertBaseColorMetalnessToAlbedoRf0 (Composition.cs.hlsl:184-185)
NRD_FE void NRD_ConvertBaseColorMetalnessToAlbedoRf0(float3_ baseColor, float metalness, float3_& albedo, float3_& Rf0) {
    albedo = {baseColor.x * (1.0f - metalness), baseColor.y * (1.0f - metalness), baseColor.z * (1.0f - metalness)};
    Rf0 = {fe_fma(baseColor.x - 0.04f, metalness, 0.04f), fe_fma(baseColor.y - 0.04f, metalness, 0.04f), fe_fma(baseColor.z - 0.04f, metalness, 0.04f)};
}
NRD_FE float NRD_SrgbToLinear(float s) { return s <= 0.04045f ? s / 12.92f : fe_pow((s + 0.055f) / 1.055f, 2.4f); }

// SH / "spherical gaussian" view of a denoised {SH0, SH1} pair. c0 = luma, c1 = sum of direction x luma.
struct NRD_SG {
    float3_ color; // SH0 colour, linear RGB
    float c0;      // its luma
    float3_ c1;
    float normHitDist;
};
NRD_FE NRD_SG REBLUR_BackEnd_UnpackSh(float4_ sh0, float3_ sh1) {
    float3_ c = NRD_YCoCgToLinear({sh0.x, sh0.y, sh0.z});
    return {c, sh0.x, sh1, sh0.w};
}
NRD_FE NRD_SG RELAX_BackEnd_UnpackSh(float4_ sh0, float3_ sh1) { return {{sh0.x, sh0.y, sh0.z}, fe_luma({sh0.x, sh0.y, sh0.z}), sh1, sh0.w}; }
NRD_FE float3_ NRD_SG_ExtractColor(NRD_SG sg) { return sg.color; }
// luma seen along `dir` relative to the stored luma: Y(dir) = 2/3 max(c0 / 2 + dir . c1, 0), so light arriving head-on keeps its luma
NRD_FE float NRD_SG_ResolveScale(NRD_SG sg, float3_ dir) {
    float Y = fe_max(fe_fma(0.5f, sg.c0, fe_dot(dir, sg.c1)), 0.0f) * (2.0f / 3.0f);
    return Y / fe_max(sg.c0, 1e-6f);
}
NRD_FE float fe_spec_dominant_factor(float roughness) {
    float s = fe_sat(1.0f - roughness);
    return s * (__builtin_sqrtf(s) + roughness);
}
NRD_FE float3_ NRD_SG_SpecularDirection(float3_ N, float3_ V, float roughness) {
    float NoV = fe_dot(N, V);
    float3_ R = {N.x * 2.0f * NoV - V.x, N.y * 2.0f * NoV - V.y, N.z * 2.0f * NoV - V.z};
    float f = fe_spec_dominant_factor(roughness);
    return fe_normalize({fe_fma(R.x - N.x, f, N.x), fe_fma(R.y - N.y, f, N.y), fe_fma(R.z - N.z, f, N.z)});
}
NRD_FE float3_ NRD_SG_ResolveDiffuse(NRD_SG sg, float3_ N, float3_, float) {
    float s = NRD_SG_ResolveScale(sg, N);
    return {sg.color.x * s, sg.color.y * s, sg.color.z * s};
}
NRD_FE float3_ NRD_SG_ResolveSpecular(NRD_SG sg, float3_ N, float3_ V, float roughness) {
    float s = NRD_SG_ResolveScale(sg, NRD_SG_SpecularDirection(N, V, roughness));
    return {sg.color.x * s, sg.color.y * s, sg.color.z * s};
}
// Re-jitter (Composition.cs.hlsl:92-107): denoising averages the signal over pixels with slightly different normals; the ratio
// of this pixel's resolve to the depth-weighted mean of the resolves against its 4 neighbours' normals puts the per-pixel
// detail back. Returns {diffuse scale, specular scale}, each in [0, 2].
NRD_FE void NRD_SG_ReJitter(NRD_SG diffSg, NRD_SG specSg, float3_ V, float roughness, float Z, const float Zn[4], float3_ N, const float3_ Nn[4], float& diffScale,
                            float& specScale) {
    float dc = NRD_SG_ResolveScale(diffSg, N), sc = NRD_SG_ResolveScale(specSg, NRD_SG_SpecularDirection(N, V, roughness));
    float dsum = dc, ssum = sc, wsum = 1.0f;
    for (int i = 0; i < 4; i++) {
        float w = fe_abs(Zn[i] - Z) <= 0.05f * fe_max(fe_abs(Z), fe_abs(Zn[i])) ? 1.0f : 0.0f;
        dsum = fe_fma(NRD_SG_ResolveScale(diffSg, Nn[i]), w, dsum);
        ssum = fe_fma(NRD_SG_ResolveScale(specSg, NRD_SG_SpecularDirection(Nn[i], V, roughness)), w, ssum);
        wsum += w;
    }
    float dm = dsum / wsum, sm = ssum / wsum;
    diffScale = dm > 1e-6f ? fe_min(dc / dm, 2.0f) : 1.0f;
    specScale = sm > 1e-6f ? fe_min(sc / sm, 2.0f) : 1.0f;
}

} // namespace nrd_fe

#endif // NRD_FRONTEND_H
