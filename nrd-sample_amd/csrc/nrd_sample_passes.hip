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

__global__ __launch_bounds__(256) void k_confidence_blur(const ConfidenceParams p) {
    int x = (int)(blockIdx.x * 16 + threadIdx.x), y = (int)(blockIdx.y * 16 + threadIdx.y);
    if (x >= p.W || y >= p.H)
        return;
    uint2 raw0 = ld<uint2>(p.in, x, y, 8);
    f4 d0 = unpack_h4(raw0);
    float z0 = d0.w / FP16_VIEWZ_SCALE;
    bool last = p.step == 5;
    if (absf(z0) > SAMPLE_INF) { // :42-46
        st<uint2>(p.out, x, y, 8, uint2{(raw0.x & 0xffff0000u) | (uint32_t)f2h(last ? 1.0f : 0.0f), raw0.y});
        return;
    }
    float u0 = ((float)x + 0.5f) * p.invW, v0 = ((float)y + 0.5f) * p.invH;
    f3 Xv0 = reconstruct_view(p.frustum, u0, v0, z0, p.ortho);
    f3 Nv0 = oct_decode(d0.y, d0.z);
    // GetGeometryWeightParams (:18-28)
    float frustumSize = p.rectW * p.unproject * lerpf(absf(Xv0.z), 1.0f, absf(p.ortho));
    float ga = 1.0f / (0.02f * frustumSize);
    float gb = -(dot3(Nv0, Xv0) * ga);
    float gradient = d0.x, sum = 1.0f;
    // all 24 gathers first (nearest, clamp-to-edge: SampleLevel( gNearestClamp ) :66), arithmetic after
    uint2 raw[24];
    int k = 0;
#pragma unroll
    for (int i = -2; i <= 2; i++)
#pragma unroll
        for (int j = -2; j <= 2; j++) {
            if (i == 0 && j == 0)
                continue;
            int px = imin(imax(x + i * p.step, 0), p.W - 1), py = imin(imax(y + j * p.step, 0), p.H - 1);
            raw[k++] = ld<uint2>(p.in, px, py, 8);
        }
    k = 0;
#pragma unroll
    for (int i = -2; i <= 2; i++)
#pragma unroll
        for (int j = -2; j <= 2; j++) {
            if (i == 0 && j == 0)
                continue;
            f4 d = unpack_h4(raw[k++]);
            float u = ((float)(x + i * p.step) + 0.5f) * p.invW, v = ((float)(y + j * p.step) + 0.5f) * p.invH;
            float w = gauss_weight(i * i + j * j);
            float z = d.w / FP16_VIEWZ_SCALE;
            f3 Xv = reconstruct_view(p.frustum, u, v, z, p.ortho);
            float NoX = dot3(Nv0, Xv);
            w *= smoothstep01(1.0f - absf(fma_(NoX, ga, gb))); // Math::SmoothStep( 1, 0, |x| ) == smoothstep( saturate( 1 - |x| ) )
            f3 Nv = oct_decode(d.y, d.z);
            float NoN = sat(dot3(Nv0, Nv));
            w *= NoN * NoN;
            gradient = fma_(d.x, w, gradient);
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
        hipLaunchKernelGGL(k_confidence_blur, grid, dim3(16, 16, 1), 0, s, p);
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
}
