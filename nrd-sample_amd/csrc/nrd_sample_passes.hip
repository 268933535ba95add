// nrd_sample_passes.hip - the sample-side image passes either side of the denoiser path (SURVEY.md 8f "next" rows) as gfx950
// HIP kernels behind the C-ABI of include/nrdhip.h:
//   nrdhip_confidence_blur  = 5 x Shaders/ConfidenceBlur.cs.hlsl:33-106 (dispatch loop Source/NRDSample.cpp:3999-4026)
//   nrdhip_backend_unpack   = the NRD-facing part of Shaders/Composition.cs.hlsl:57-64, :74-175
// ConfidenceBlur.cs.hlsl IS in the reference tree and is restated statement by statement; the MathLib helpers it calls
// (Math::SmoothStep, Geometry::ReconstructViewPosition, Packing::DecodeUnitVector, Color::HdrToLinear_Uncharted, Color::ToSrgb,
// Sequence::Bayer4x4 - "ml.hlsli", an absent submodule) are restated from their published definitions [RECOLLECTION] and frozen
// in oracle/orc_sample_passes.cpp. Both kernels are streaming / small-stencil filters: HBM-bound, no LDS, no MFMA.
#include "nrd_device.h"

#include "../../include/nrdhip.h"

namespace nrdhip {
namespace {

constexpr float FP16_VIEWZ_SCALE = 0.125f; // Shaders/Shared.hlsli:143
constexpr float SAMPLE_INF = 1e5f;         // Shaders/Shared.hlsli:141

struct ConfidenceParams {
    PlaneRef in, out;
    int W, H, step;
    float frustum[4], invW, invH, rectW, unproject, ortho;
    uint32_t frameIndex;
    float maxAccum;
    int relax;
};

// Geometry::ReconstructViewPosition( uv, cameraFrustum, viewZ, orthoMode )
NRD_DEV f3 reconstruct_view(const float* fr, float u, float v, float z, float ortho) {
    float s = fma_(z, 1.0f - absf(ortho), ortho);
    return {fma_(u, fr[2], fr[0]) * s, fma_(v, fr[3], fr[1]) * s, z};
}

// Color::HdrToLinear_Uncharted (Hable filmic curve, white point 11.2), Color::ToSrgb
NRD_DEV float uncharted_curve(float x) {
    const float A = 0.22f, B = 0.3f, C = 0.1f, D = 0.2f, E = 0.01f, F = 0.3f;
    return fma_(x, fma_(A, x, C * B), D * E) / fma_(x, fma_(A, x, B), D * F) - E / F;
}
NRD_DEV float hdr_to_linear_uncharted(float x) { return uncharted_curve(x) / uncharted_curve(11.2f); }
NRD_DEV float to_srgb(float x) { return x < 0.0031308f ? 12.92f * x : fma_(1.055f, pow01(x, 1.0f / 2.4f), -0.055f); }

// Sequence::Bayer4x4( pixelPos, frameIndex )
NRD_DEV float bayer4x4(uint32_t x, uint32_t y, uint32_t frameIndex) {
    uint32_t wx = x & 3u, wy = y & 3u;
    uint32_t a = 2068378560u * (1u - (wx >> 1)) + 1500172770u * (wx >> 1);
    uint32_t b = (wy + ((wx & 1u) << 2)) << 2;
    return (float)(((a >> b) + frameIndex) & 0xFu) * 0.0625f;
}

// exp( -2 d^2 ), d = length( int2( i, j ) ) / 2  ->  exp( -( i^2 + j^2 ) / 2 ), indexed by i^2 + j^2 (ConfidenceBlur.cs.hlsl:69-70)
NRD_DEV float gauss_weight(int r2) {
    return r2 == 1 ? 0.60653066f : r2 == 2 ? 0.36787944f : r2 == 4 ? 0.13533528f : r2 == 5 ? 0.082084999f : 0.018315639f;
}

// One workgroup = one 16 x 16 tile. Every texel of the (16 + 4 STEP)^2 window around it is read by up to 24 pixels of the tile, and what a tap
// needs from it - the gradient, the view position of the TAP (unclamped uv, the clamped texel's depth: SampleLevel( gNearestClamp ) :66) and
// the decoded normal - does not depend on which pixel asks: it is decoded ONCE per window position into LDS (one array per component:
// neighbouring lanes read neighbouring words), and a tap is 7 LDS reads + ~18 instructions instead of a gather + two conversions + an
// octahedron decode with its normalisation + a view-position reconstruction (1625 -> ~700 VALU instructions per pixel; this pass is five
// dependent launches over 1 / 25 of the frame's pixels, so what it costs is the instructions of a wave, profiles/r05_ab_confidence_blur.txt).
// The values are the per-tap code's, expression by expression.
template <int STEP>
__global__ __launch_bounds__(256) void k_confidence_blur(const ConfidenceParams p) {
    constexpr int T = 16 + 4 * STEP, N = T * T;
    __shared__ float sG[N], sXx[N], sXy[N], sXz[N], sNx[N], sNy[N], sNz[N];
    const int tid = (int)threadIdx.y * 16 + (int)threadIdx.x;
    const int x0 = (int)blockIdx.x * 16 - 2 * STEP, y0 = (int)blockIdx.y * 16 - 2 * STEP;
    constexpr int TRIPS = (N + 255) / 256;
    uint2 raw[TRIPS];
#pragma unroll
    for (int k = 0; k < TRIPS; k++) { // all loads of the thread first
        const int i = imin(tid + 256 * k, N - 1);
        const int ly = i / T, lx = i - ly * T;
        const int px = imin(imax(x0 + lx, 0), p.W - 1), py = imin(imax(y0 + ly, 0), p.H - 1);
        raw[k] = ld<uint2>(p.in, px, py, 8);
    }
#pragma unroll
    for (int k = 0; k < TRIPS; k++) {
        const int i = tid + 256 * k;
        if (i < N) {
            const int ly = i / T, lx = i - ly * T;
            const f4 d = unpack_h4(raw[k]);
            const float u = ((float)(x0 + lx) + 0.5f) * p.invW, v = ((float)(y0 + ly) + 0.5f) * p.invH;
            const float z = d.w / FP16_VIEWZ_SCALE;
            const f3 Xv = reconstruct_view(p.frustum, u, v, z, p.ortho);
            const f3 Nv = oct_decode(d.y, d.z);
            sG[i] = d.x;
            sXx[i] = Xv.x;
            sXy[i] = Xv.y;
            sXz[i] = Xv.z;
            sNx[i] = Nv.x;
            sNy[i] = Nv.y;
            sNz[i] = Nv.z;
        }
    }
    __syncthreads();
    const int x = (int)(blockIdx.x * 16 + threadIdx.x), y = (int)(blockIdx.y * 16 + threadIdx.y);
    if (x >= p.W || y >= p.H)
        return;
    const uint2 raw0 = ld<uint2>(p.in, x, y, 8); // (the texel itself: its words travel on to the output)
    const int ci = ((int)threadIdx.y + 2 * STEP) * T + (int)threadIdx.x + 2 * STEP;
    const f3 Xv0 = {sXx[ci], sXy[ci], sXz[ci]};
    const float z0 = Xv0.z;
    const bool last = STEP == 5;
    if (absf(z0) > SAMPLE_INF) { // :42-46
        st<uint2>(p.out, x, y, 8, uint2{(raw0.x & 0xffff0000u) | (uint32_t)f2h(last ? 1.0f : 0.0f), raw0.y});
        return;
    }
    const f3 Nv0 = {sNx[ci], sNy[ci], sNz[ci]};
    // GetGeometryWeightParams (:18-28)
    float frustumSize = p.rectW * p.unproject * lerpf(absf(Xv0.z), 1.0f, absf(p.ortho));
    float ga = 1.0f / (0.02f * frustumSize);
    float gb = -(dot3(Nv0, Xv0) * ga);
    float gradient = sG[ci], sum = 1.0f;
#pragma unroll
    for (int i = -2; i <= 2; i++)
#pragma unroll
        for (int j = -2; j <= 2; j++) {
            if (i == 0 && j == 0)
                continue;
            const int q = ci + (j * T + i) * STEP;
            float w = gauss_weight(i * i + j * j);
            const f3 Xv = {sXx[q], sXy[q], sXz[q]};
            float NoX = dot3(Nv0, Xv);
            w *= smoothstep01(1.0f - absf(fma_(NoX, ga, gb))); // Math::SmoothStep( 1, 0, |x| ) == smoothstep( saturate( 1 - |x| ) )
            const f3 Nv = {sNx[q], sNy[q], sNz[q]};
            float NoN = sat(dot3(Nv0, Nv));
            w *= NoN * NoN;
            gradient = fma_(sG[q], w, gradient);
            sum += w;
        }
    gradient /= sum;
    if (last) { // :86-103 "gradient" -> "history confidence"
        gradient = hdr_to_linear_uncharted(gradient);
        gradient = 1.0f - to_srgb(sat(gradient));
        if (p.relax)
            gradient *= gradient;
        float dither = bayer4x4((uint32_t)x, (uint32_t)y, p.frameIndex);
        gradient += (dither - 0.5f) / p.maxAccum;
    }
    st<uint2>(p.out, x, y, 8, uint2{(raw0.x & 0xffff0000u) | (uint32_t)f2h(sat(gradient)), raw0.y});
}

struct UnpackParams {
    int W, H, mode, relax, resolve, shadowBpt;
    PlaneRef diff, spec, diff1, spec1, nr, shadow, outDiff, outSpec, outShadow;
    float v2w[9], frustum[4], invW, invH;
};

// SH resolve of this build's SH encoding (SH0 = {Y, Co, Cg, hitT}, SH1.xyz = sum of direction * Y): luminance seen along `dir`,
// normalised so that light arriving head-on keeps its luminance; returns the scale to apply to the extracted colour
NRD_DEV float sh_resolve_scale(float c0, f3 c1, f3 dir) {
    float Y = fmax2(fma_(0.5f, c0, dot3(dir, c1)), 0.0f) * (2.0f / 3.0f);
    return Y / fmax2(c0, 1e-6f);
}

__global__ __launch_bounds__(256) void k_backend_unpack(const UnpackParams p) {
    int x = (int)(blockIdx.x * 64 + threadIdx.x), y = (int)(blockIdx.y * 4 + threadIdx.y);
    if (x >= p.W || y >= p.H)
        return;
    if (p.outShadow.p) { // SIGMA_BackEnd_UnpackShadow (Composition.cs.hlsl:57-64): stored sqrt-encoded
        f4 s;
        if (p.shadowBpt == 1) {
            float v = (float)ld<uint8_t>(p.shadow, x, y, 1) * (1.0f / 255.0f);
            s = {v, v, v, v};
        } else {
            uint32_t r = ld<uint32_t>(p.shadow, x, y, 4);
            s = {(float)(r & 255u) * (1.0f / 255.0f), (float)((r >> 8) & 255u) * (1.0f / 255.0f), (float)((r >> 16) & 255u) * (1.0f / 255.0f),
                 (float)(r >> 24) * (1.0f / 255.0f)};
        }
        st<uint2>(p.outShadow, x, y, 8, pack_h4({s.x * s.x, s.y * s.y, s.z * s.z, s.w * s.w}));
    }
    f3 N = {0.0f, 0.0f, 1.0f}, V = {0.0f, 0.0f, 1.0f};
    float roughness = 1.0f;
    const bool sh = p.mode == NRDHIP_UNPACK_SH;
    if (sh && p.resolve) {
        uint32_t nr = ld<uint32_t>(p.nr, x, y, 4);
        N = oct_decode((float)(nr & 1023u) * (1.0f / 1023.0f), (float)((nr >> 10) & 1023u) * (1.0f / 1023.0f));
        roughness = (float)((nr >> 20) & 1023u) * (1.0f / 1023.0f);
        float u = ((float)x + 0.5f) * p.invW, v = ((float)y + 0.5f) * p.invH;
        f3 Xv = {fma_(u, p.frustum[2], p.frustum[0]), fma_(v, p.frustum[3], p.frustum[1]), 1.0f};
        V = normalize3(rot3(p.v2w, mul3(Xv, -1.0f)));
    }
#pragma unroll
    for (int sig = 0; sig < 2; sig++) {
        const PlaneRef& in = sig ? p.spec : p.diff;
        const PlaneRef& in1 = sig ? p.spec1 : p.diff1;
        const PlaneRef& out = sig ? p.outSpec : p.outDiff;
        if (!in.p || !out.p)
            continue;
        f4 o;
        if (p.mode == NRDHIP_UNPACK_OCCLUSION) { // Composition.cs.hlsl:124-126
            float h = (float)ld<uint16_t>(in, x, y, 2) * (1.0f / 65535.0f);
            o = {h, h, h, h};
        } else {
            f4 v = unpack_h4(ld<uint2>(in, x, y, 8));
            f3 rgb = p.relax ? f3{v.x, v.y, v.z} : ycocg_to_linear({v.x, v.y, v.z}); // RELAX_BackEnd_UnpackRadiance / REBLUR_BackEnd_Unpack...
            if (!p.relax)
                rgb = {fmax2(rgb.x, 0.0f), fmax2(rgb.y, 0.0f), fmax2(rgb.z, 0.0f)};
            if (sh && p.resolve) { // :85-122 without the re-jitter step
                f4 v1 = unpack_h4(ld<uint2>(in1, x, y, 8));
                f3 dir = N;
                if (sig) {
                    float NoV = dot3(N, V);
                    f3 R = sub3(mul3(N, 2.0f * NoV), V);
                    dir = normalize3(add3(N, mul3(sub3(R, N), spec_dominant_factor(roughness))));
                }
                float Y = p.relax ? fma_(0.25f, v.x, fma_(0.5f, v.y, 0.25f * v.z)) : v.x;
                rgb = mul3(rgb, sh_resolve_scale(Y, {v1.x, v1.y, v1.z}, dir));
            }
            o = {rgb.x, rgb.y, rgb.z, p.relax ? 0.318309886f : v.w}; // RELAX has no AO / SO: 1 / pi (:176-181)
        }
        st<uint2>(out, x, y, 8, pack_h4(o));
    }
}

// =====================================================================================================================
// TAA (Shaders/Taa.cs.hlsl:11-159). Helpers outside that file: ApplyTonemap and BicubicFilterNoCorners are in-tree
// (Shaders/Shared.hlsli:337-387); Color::ClampAabb, Color::RgbToXyz, Math::PositiveRcp are MathLib [RECOLLECTION].
// =====================================================================================================================
struct TaaParams {
    PlaneRef mv, composed, history, result;
    int W, H, RW, RH, tonemap;
    float Wp, Hp, invW, invH, invRW, invRH, hdrScale, taa;
};

NRD_DEV float positive_rcp(float x) { return rcp_(fmax2(x, 1e-15f)); }
NRD_DEV f3 taa_tonemap(const TaaParams& p, f3 c) { // ApplyTonemap (Shared.hlsli:337-347)
    if (!p.tonemap)
        return c;
    return {p.hdrScale * hdr_to_linear_uncharted(c.x), p.hdrScale * hdr_to_linear_uncharted(c.y), p.hdrScale * hdr_to_linear_uncharted(c.z)};
}
// SampleLevel( gLinearClamp, uv, 0 ) of an RGBA16F texture
NRD_DEV f4 sample_linear_clamp(const PlaneRef& P, float u, float v, int w, int h) {
    float x = fma_(u, (float)w, -0.5f), y = fma_(v, (float)h, -0.5f);
    float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y);
    float fx = x - x0, fy = y - y0;
    x0 = clampf(x0, -1.0f, (float)w);
    y0 = clampf(y0, -1.0f, (float)h);
    int ix0 = imin(imax((int)x0, 0), w - 1), ix1 = imin(imax((int)x0 + 1, 0), w - 1);
    int iy0 = imin(imax((int)y0, 0), h - 1), iy1 = imin(imax((int)y0 + 1, 0), h - 1);
    // lerp(a, b, t) = fma(b - a, t, a) on fp16 texels: the difference is ONE v_fma_mix_f32 that reads both halves as they are (hdiff_,
    // nrd_device.h: the same single rounding as the subtraction of the converted values), and the blend a second one with a as its fp16
    // addend: 2 instructions per channel where convert, convert, subtract, fma took 4 - 16 channels x 5 samples per pixel
    const uint2 ta = ld<uint2>(P, ix0, iy0, 8), tb = ld<uint2>(P, ix1, iy0, 8), tc = ld<uint2>(P, ix0, iy1, 8), td = ld<uint2>(P, ix1, iy1, 8);
    float r[4];
    {
        const float t0 = fma_(hdiff_<false, false>(tb.x, ta.x), fx, h2f((uint16_t)(ta.x & 0xffffu))), b0 = fma_(hdiff_<false, false>(td.x, tc.x), fx, h2f((uint16_t)(tc.x & 0xffffu)));
        const float t1 = fma_(hdiff_<true, true>(tb.x, ta.x), fx, h2f((uint16_t)(ta.x >> 16))), b1 = fma_(hdiff_<true, true>(td.x, tc.x), fx, h2f((uint16_t)(tc.x >> 16)));
        const float t2 = fma_(hdiff_<false, false>(tb.y, ta.y), fx, h2f((uint16_t)(ta.y & 0xffffu))), b2 = fma_(hdiff_<false, false>(td.y, tc.y), fx, h2f((uint16_t)(tc.y & 0xffffu)));
        const float t3 = fma_(hdiff_<true, true>(tb.y, ta.y), fx, h2f((uint16_t)(ta.y >> 16))), b3 = fma_(hdiff_<true, true>(td.y, tc.y), fx, h2f((uint16_t)(tc.y >> 16)));
        r[0] = lerpf(t0, b0, fy);
        r[1] = lerpf(t1, b1, fy);
        r[2] = lerpf(t2, b2, fy);
        r[3] = lerpf(t3, b3, fy);
    }
    return {r[0], r[1], r[2], r[3]};
}
// BicubicFilterNoCorners (Shared.hlsli:349-387)
NRD_DEV f4 bicubic_no_corners(const TaaParams& p, float sx, float sy) {
    const float sh = 0.66f; // TAA_HISTORY_SHARPNESS (Shared.hlsli:148)
    float cx = __builtin_floorf(sx - 0.5f) + 0.5f, cy = __builtin_floorf(sy - 0.5f) + 0.5f;
    float f[2] = {sat(sx - cx), sat(sy - cy)};
    float w0[2], w3[2], wl2[2], tc2[2];
    const float c[2] = {cx, cy}, inv[2] = {p.invRW, p.invRH};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        float f1 = f[k], f2 = f1 * f1, f3v = f1 * f2;
        w0[k] = -sh * f3v + 2.0f * sh * f2 - sh * f1;
        float w1 = (2.0f - sh) * f3v - (3.0f - sh) * f2 + 1.0f;
        float w2 = -(2.0f - sh) * f3v + (3.0f - 2.0f * sh) * f2 + sh * f1;
        w3[k] = sh * f3v - sh * f2;
        wl2[k] = w1 + w2;
        tc2[k] = inv[k] * (c[k] + w2 * positive_rcp(wl2[k]));
    }
    float tc0x = p.invRW * (cx - 1.0f), tc0y = p.invRH * (cy - 1.0f), tc3x = p.invRW * (cx + 2.0f), tc3y = p.invRH * (cy + 2.0f);
    float w = wl2[0] * w0[1];
    f4 color = mul4(sample_linear_clamp(p.history, tc2[0], tc0y, p.RW, p.RH), w);
    float sum = w;
    w = w0[0] * wl2[1];
    color = fma4(sample_linear_clamp(p.history, tc0x, tc2[1], p.RW, p.RH), w, color);
    sum += w;
    w = wl2[0] * wl2[1];
    color = fma4(sample_linear_clamp(p.history, tc2[0], tc2[1], p.RW, p.RH), w, color);
    sum += w;
    w = w3[0] * wl2[1];
    color = fma4(sample_linear_clamp(p.history, tc3x, tc2[1], p.RW, p.RH), w, color);
    sum += w;
    w = wl2[0] * w3[1];
    color = fma4(sample_linear_clamp(p.history, tc2[0], tc3y, p.RW, p.RH), w, color);
    sum += w;
    return mul4(color, positive_rcp(sum));
}
NRD_DEV f3 rgb_to_xyz(f3 c) { // Color::RgbToXyz (sRGB primaries, D65, Y in 0..100)
    return {100.0f * fma_(0.1804808f, c.z, fma_(0.3575843f, c.y, 0.4123908f * c.x)), 100.0f * fma_(0.0721923f, c.z, fma_(0.7151687f, c.y, 0.2126390f * c.x)),
            100.0f * fma_(0.9505322f, c.z, fma_(0.1191948f, c.y, 0.0193308f * c.x))};
}
NRD_DEV f3 xyz_to_lab(f3 x) { // Taa.cs.hlsl:43-54; pow(x, 0.333333) = cbrt_pos_ (nrd_device.h, ledger row 19)
    x = {x.x * (1.0f / 95.0489f), x.y * (1.0f / 100.0f), x.z * (1.0f / 108.8840f)};
    float fx = x.x > 0.008856f ? cbrt_pos_(x.x) : fma_(7.787f, x.x, 16.0f / 116.0f);
    float fy = x.y > 0.008856f ? cbrt_pos_(x.y) : fma_(7.787f, x.y, 16.0f / 116.0f);
    float fz = x.z > 0.008856f ? cbrt_pos_(x.z) : fma_(7.787f, x.z, 16.0f / 116.0f);
    // NOTE: as in the shader, "l" is computed from the ALREADY transformed y
    float l = fy > 0.008856f ? fma_(116.0f, cbrt_pos_(fy), -16.0f) : 903.3f * fy;
    return {l, 500.0f * (fx - fy), 200.0f * (fy - fz)};
}
NRD_DEV f3 clamp_aabb(f3 center, f3 ext, f3 prev, bool& moved) { // Color::ClampAabb: clip towards the box centre
    f3 d = sub3(prev, center);
    f3 dn = {absf(d.x * positive_rcp(ext.x)), absf(d.y * positive_rcp(ext.y)), absf(d.z * positive_rcp(ext.z))};
    float maxd = fmax2(dn.x, fmax2(dn.y, dn.z));
    float r = positive_rcp(maxd);
    f3 t = {fma_(d.x, r, center.x), fma_(d.y, r, center.y), fma_(d.z, r, center.z)};
    moved = maxd > 1.0f;
    return moved ? t : prev;
}

__global__ __launch_bounds__(256) void k_taa(const TaaParams p) {
    // one 16-byte LDS texel per window position: {tonemapped colour, signed viewZ} - what each of the 25 (or 9) moment taps reads, in
    // ONE ds_read_b128 (the transliterated form kept six float arrays: four scalar reads per tap; 0.244 ms at 4K,
    // profiles/r03v2_bench_sample_passes.json); the motion xy is only read at the chosen offset and keeps its own array
    __shared__ float4 sTile[400];
    __shared__ float2 sMvXY[400];
    const int tidx = (int)threadIdx.x, tidy = (int)threadIdx.y;
    const int x = (int)blockIdx.x * 16 + tidx, y = (int)blockIdx.y * 16 + tidy;
    {   // PRELOAD_INTO_SMEM (:17-28, :33-39)
        int bx = (int)blockIdx.x * 16 - 2, by = (int)blockIdx.y * 16 - 2;
        for (int i = tidy * 16 + tidx; i < 400; i += 256) {
            const int ly = (i * 3277) >> 16, lx = i - ly * 20; // i / 20 for i < 400 without a division
            int gx = imin(imax(bx + lx, 0), p.W - 1), gy = imin(imax(by + ly, 0), p.H - 1);
            f4 c = unpack_h4(ld<uint2>(p.composed, gx, gy, 8));
            f3 t = taa_tonemap(p, {c.x, c.y, c.z});
            f4 m = unpack_h4(ld<uint2>(p.mv, gx, gy, 8));
            sTile[i] = float4{t.x, t.y, t.z, m.w}; // dZ is not needed
            sMvXY[i] = float2{m.x, m.y};
        }
    }
    __syncthreads();
    float u = ((float)x + 0.5f) * p.invW, v = ((float)y + 0.5f) * p.invH;
    if (u > 1.0f || v > 1.0f)
        return;
    float sum = 0.0f;
    f3 m1 = {0, 0, 0}, m2 = {0, 0, 0}, input = {0, 0, 0};
    const int ci = (tidy + 2) * 20 + tidx + 2;
    float centerZ = sTile[ci].w;
    float minViewZ = absf(centerZ);
    int mi = ci; // window index of the closest-depth neighbour (offseti of the shader, :72, :94-98)
    const bool want5x5 = centerZ < 0.0f;
#pragma unroll
    for (int dy = 0; dy <= 4; dy++)
#pragma unroll
        for (int dx = 0; dx <= 4; dx++) {
            const bool border = dx == 0 || dx == 4 || dy == 0 || dy == 4;
            if (border && !want5x5)
                continue;
            int si = (tidy + dy) * 20 + tidx + dx;
            const float4 tt = sTile[si];
            f3 c = {tt.x, tt.y, tt.z};
            float viewZ = absf(tt.w);
            if (dx == 2 && dy == 2)
                input = c;
            else if (viewZ < minViewZ) {
                minViewZ = viewZ;
                mi = si;
            }
            // r2 = LengthSquared( offset / BORDER - 1.0 ) with the INTEGER division of the shader (:103): -1,-1,0,0,1 per axis
            const int qx = dx / 2 - 1, qy = dy / 2 - 1;
            const int r2 = qx * qx + qy * qy;
            const float w = r2 == 0 ? 1.0f : (r2 == 1 ? 0.36787944f : 0.13533528f); // exp( -r2 )
            m1 = {fma_(c.x, w, m1.x), fma_(c.y, w, m1.y), fma_(c.z, w, m1.z)};
            m2 = {fma_(c.x * c.x, w, m2.x), fma_(c.y * c.y, w, m2.y), fma_(c.z * c.z, w, m2.z)};
            sum += w;
        }
    float rs = rcp_(sum);
    m1 = mul3(m1, rs);
    m2 = mul3(m2, rs);
    f3 sigma = {sqrt_(absf(m2.x - m1.x * m1.x)) * 2.0f, sqrt_(absf(m2.y - m1.y * m1.y)) * 2.0f, sqrt_(absf(m2.z - m1.z * m1.z)) * 2.0f}; // TAA_SIGMA_SCALE
    // previous pixel position (:118-119): motion of the closest-depth neighbour
    const float2 mxy = sMvXY[mi];
    float pu = fma_(mxy.x, p.invW, u), pv = fma_(mxy.y, p.invH, v);
    f4 history = bicubic_no_corners(p, sat(pu) * p.Wp, sat(pv) * p.Hp);
    f3 hist = {fmax2(history.x, 0.0f), fmax2(history.y, 0.0f), fmax2(history.z, 0.0f)};
    float mixRate = sat(history.w);
    mixRate = mixRate * rcp_(1.0f + mixRate);
    bool inScreen = sat(pu) == pu && sat(pv) == pv;
    mixRate = inScreen ? mixRate : 1.0f;
    bool moved;
    f3 clamped = clamp_aabb(m1, sigma, hist, moved);
    // Disocclusion #2: the Lab distance between the history and its clamped copy. A history inside the box comes back unchanged - the two
    // conversions then return the same bits and the distance is an exact 0 (sqrt_(0) = 0): the eight cube roots are skipped for it
    // (a wave of a converged image skips them altogether)
    float diff = 0.0f;
    if (moved) {
        f3 a = xyz_to_lab(rgb_to_xyz(clamped)), b = xyz_to_lab(rgb_to_xyz(hist));
        f3 dl = sub3(a, b);
        diff = sqrt_(dot3(dl, dl)) * (1.0f / (2.3f * 3.0f)); // JND = 2.3
    }
    mixRate = sat(mixRate + diff);
    float t = fmax2(mixRate, p.taa);
    f3 r = {lerpf(clamped.x, input.x, t), lerpf(clamped.y, input.y, t), lerpf(clamped.z, input.z, t)};
    st<uint2>(p.result, x, y, 8, pack_h4({r.x, r.y, r.z, mixRate}));
}

PlaneRef plane(const void* p, uint32_t pitch, uint16_t w, uint16_t h) { return PlaneRef{(uint8_t*)p, pitch, w, h}; }

} // namespace
} // namespace nrdhip

extern "C" {

NRDHIP_API int nrdhip_confidence_blur(const nrdhip_confidence_blur_desc* d, void* hip_stream) {
    using namespace nrdhip;
    if (!d || !d->ping || !d->pong || !d->width || !d->height || d->pitch_bytes < (uint32_t)d->width * 8u || d->first_pass + d->passes_num > 5u)
        return 2; // nrd::Result::INVALID_ARGUMENT
    hipStream_t s = (hipStream_t)hip_stream;
    ConfidenceParams p = {};
    p.W = d->width;
    p.H = d->height;
    for (int i = 0; i < 4; i++)
        p.frustum[i] = d->camera_frustum[i];
    p.invW = d->inv_size[0];
    p.invH = d->inv_size[1];
    p.rectW = d->rect_width;
    p.unproject = d->unproject;
    p.ortho = d->ortho_mode;
    p.frameIndex = d->frame_index;
    p.maxAccum = (float)d->max_accumulated_frame_num;
    p.relax = d->relax ? 1 : 0;
    dim3 grid((unsigned)((d->width + 15) / 16), (unsigned)((d->height + 15) / 16), 1);
    for (uint32_t i = d->first_pass; i < d->first_pass + d->passes_num; i++) {
        bool even = (i & 1u) == 0u; // Source/NRDSample.cpp:4004-4008
        p.in = plane(even ? d->ping : d->pong, d->pitch_bytes, d->width, d->height);
        p.out = plane(even ? d->pong : d->ping, d->pitch_bytes, d->width, d->height);
        p.step = (int)(1u + i);
        switch (p.step) {
        case 1: hipLaunchKernelGGL(k_confidence_blur<1>, grid, dim3(16, 16, 1), 0, s, p); break;
        case 2: hipLaunchKernelGGL(k_confidence_blur<2>, grid, dim3(16, 16, 1), 0, s, p); break;
        case 3: hipLaunchKernelGGL(k_confidence_blur<3>, grid, dim3(16, 16, 1), 0, s, p); break;
        case 4: hipLaunchKernelGGL(k_confidence_blur<4>, grid, dim3(16, 16, 1), 0, s, p); break;
        default: hipLaunchKernelGGL(k_confidence_blur<5>, grid, dim3(16, 16, 1), 0, s, p); break;
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

NRDHIP_API int nrdhip_backend_unpack(const nrdhip_unpack_desc* d, void* hip_stream) {
    using namespace nrdhip;
    if (!d || !d->width || !d->height || d->mode > NRDHIP_UNPACK_SH)
        return 2;
    if (d->mode == NRDHIP_UNPACK_SH && ((d->diff && d->out_diff && !d->diff_sh1 && d->resolve) || (d->spec && d->out_spec && !d->spec_sh1 && d->resolve) ||
                                        (d->resolve && !d->normal_roughness)))
        return 2;
    if (d->out_shadow && (!d->shadow || (d->shadow_bytes_per_texel != 1 && d->shadow_bytes_per_texel != 4)))
        return 2;
    UnpackParams p = {};
    p.W = d->width;
    p.H = d->height;
    p.mode = (int)d->mode;
    p.relax = d->relax ? 1 : 0;
    p.resolve = d->resolve ? 1 : 0;
    p.shadowBpt = (int)d->shadow_bytes_per_texel;
    p.diff = plane(d->diff, d->diff_pitch, d->width, d->height);
    p.spec = plane(d->spec, d->spec_pitch, d->width, d->height);
    p.diff1 = plane(d->diff_sh1, d->diff_sh1_pitch, d->width, d->height);
    p.spec1 = plane(d->spec_sh1, d->spec_sh1_pitch, d->width, d->height);
    p.nr = plane(d->normal_roughness, d->normal_roughness_pitch, d->width, d->height);
    p.shadow = plane(d->shadow, d->shadow_pitch, d->width, d->height);
    p.outDiff = plane(d->out_diff, d->out_diff_pitch, d->width, d->height);
    p.outSpec = plane(d->out_spec, d->out_spec_pitch, d->width, d->height);
    p.outShadow = plane(d->out_shadow, d->out_shadow_pitch, d->width, d->height);
    for (int i = 0; i < 9; i++)
        p.v2w[i] = d->view_to_world[i];
    for (int i = 0; i < 4; i++)
        p.frustum[i] = d->camera_frustum[i];
    p.invW = d->inv_rect_size[0];
    p.invH = d->inv_rect_size[1];
    dim3 grid((unsigned)((d->width + 63) / 64), (unsigned)((d->height + 3) / 4), 1);
    hipLaunchKernelGGL(k_backend_unpack, grid, dim3(64, 4, 1), 0, (hipStream_t)hip_stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

NRDHIP_API int nrdhip_taa(const nrdhip_taa_desc* d, void* hip_stream) {
    using namespace nrdhip;
    if (!d || !d->mv || !d->composed || !d->history || !d->result || !d->rect_width || !d->rect_height || !d->render_width || !d->render_height ||
        d->rect_width > d->render_width || d->rect_height > d->render_height)
        return 2;
    TaaParams p = {};
    p.mv = plane(d->mv, d->mv_pitch, d->rect_width, d->rect_height);
    p.composed = plane(d->composed, d->composed_pitch, d->rect_width, d->rect_height);
    p.history = plane(d->history, d->history_pitch, d->render_width, d->render_height);
    p.result = plane(d->result, d->result_pitch, d->rect_width, d->rect_height);
    p.W = d->rect_width;
    p.H = d->rect_height;
    p.RW = d->render_width;
    p.RH = d->render_height;
    p.Wp = (float)(d->rect_width_prev ? d->rect_width_prev : d->rect_width);
    p.Hp = (float)(d->rect_height_prev ? d->rect_height_prev : d->rect_height);
    p.invW = 1.0f / (float)p.W;
    p.invH = 1.0f / (float)p.H;
    p.invRW = 1.0f / (float)p.RW;
    p.invRH = 1.0f / (float)p.RH;
    p.tonemap = d->tonemap ? 1 : 0;
    p.hdrScale = d->hdr_scale;
    p.taa = d->taa;
    dim3 grid((unsigned)((d->rect_width + 15) / 16), (unsigned)((d->rect_height + 15) / 16), 1);
    hipLaunchKernelGGL(k_taa, grid, dim3(16, 16, 1), 0, (hipStream_t)hip_stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
}
