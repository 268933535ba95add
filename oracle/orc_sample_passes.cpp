// orc_sample_passes.cpp - TEST INFRASTRUCTURE (CPU oracle): scalar restatement of the sample-side passes either side of the
// denoiser. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
//
//   orc_confidence_blur  follows Shaders/ConfidenceBlur.cs.hlsl:18-106 statement by statement (that file IS in the reference
//                        tree); dispatch order per Source/NRDSample.cpp:3999-4026. The MathLib helpers it calls live in an
//                        absent submodule ("ml.hlsli") and are restated from their published definitions [RECOLLECTION]:
//                        Math::SmoothStep, Geometry::ReconstructViewPosition, Packing::DecodeUnitVector (octahedral),
//                        Color::HdrToLinear_Uncharted (Hable curve, white 11.2), Color::ToSrgb, Sequence::Bayer4x4.
//                        Parity status: structure pinned to the in-tree shader, helper numerics unpinned.
//   orc_backend_unpack   the NRD-facing part of Shaders/Composition.cs.hlsl:57-64, :74-175; the NRD_SG / *_BackEnd_* helpers are
//                        NRD.hlsli (absent): decodings follow this build's own frozen encodings (oracle/README.md) - unpinned.
#include "orc_core.h"

#include "../include/nrdhip.h"

using namespace orc;

namespace {

const float FP16_VIEWZ_SCALE = 0.125f; // Shaders/Shared.hlsli:143
const float SAMPLE_INF = 1e5f;         // Shaders/Shared.hlsli:141

struct H4 {
    uint16_t v[4];
};
inline uint16_t to_h(float v) { return f32_to_f16(clampf(v, -FP16_MAX, FP16_MAX)); } // == the kernels' f2h
inline H4* texel(void* base, uint32_t pitch, int x, int y) { return (H4*)((uint8_t*)base + (size_t)y * pitch + (size_t)x * 8); }
inline const H4* texel(const void* base, uint32_t pitch, int x, int y) { return (const H4*)((const uint8_t*)base + (size_t)y * pitch + (size_t)x * 8); }

// Geometry::ReconstructViewPosition
inline f3 reconstruct_view(const float* fr, float u, float v, float z, float ortho) {
    float s = fma_(z, 1.0f - absf(ortho), ortho);
    return {fma_(u, fr[2], fr[0]) * s, fma_(v, fr[3], fr[1]) * s, z};
}
inline float uncharted_curve(float x) {
    const float A = 0.22f, B = 0.3f, C = 0.1f, D = 0.2f, E = 0.01f, F = 0.3f;
    return fma_(x, fma_(A, x, C * B), D * E) / fma_(x, fma_(A, x, B), D * F) - E / F;
}
inline float hdr_to_linear_uncharted(float x) { return uncharted_curve(x) / uncharted_curve(11.2f); }
inline float to_srgb(float x) { return x < 0.0031308f ? 12.92f * x : fma_(1.055f, pow01(x, 1.0f / 2.4f), -0.055f); }
inline float bayer4x4(uint32_t x, uint32_t y, uint32_t frameIndex) {
    uint32_t wx = x & 3u, wy = y & 3u;
    uint32_t a = 2068378560u * (1u - (wx >> 1)) + 1500172770u * (wx >> 1);
    uint32_t b = (wy + ((wx & 1u) << 2)) << 2;
    return (float)(((a >> b) + frameIndex) & 0xFu) * 0.0625f;
}
inline float gauss_weight(int r2) {
    return r2 == 1 ? 0.60653066f : r2 == 2 ? 0.36787944f : r2 == 4 ? 0.13533528f : r2 == 5 ? 0.082084999f : 0.018315639f;
}

void confidence_pass(const nrdhip_confidence_blur_desc& d, const void* in, void* out, int step) {
    const int W = d.width, H = d.height;
    const bool last = step == 5; // ConfidenceBlur.cs.hlsl:40
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const H4* t0 = texel(in, d.pitch_bytes, x, y);
            H4 o = *t0;
            float g0 = f16_to_f32(t0->v[0]), ny0 = f16_to_f32(t0->v[1]), nz0 = f16_to_f32(t0->v[2]);
            float z0 = f16_to_f32(t0->v[3]) / FP16_VIEWZ_SCALE;
            if (absf(z0) > SAMPLE_INF) { // :42-46
                o.v[0] = to_h(last ? 1.0f : 0.0f);
                *texel(out, d.pitch_bytes, x, y) = o;
                continue;
            }
            float u0 = ((float)x + 0.5f) * d.inv_size[0], v0 = ((float)y + 0.5f) * d.inv_size[1];
            f3 Xv0 = reconstruct_view(d.camera_frustum, u0, v0, z0, d.ortho_mode);
            f3 Nv0 = oct_decode({ny0, nz0});
            float frustumSize = d.rect_width * d.unproject * lerpf(absf(Xv0.z), 1.0f, absf(d.ortho_mode)); // :22
            float ga = 1.0f / (0.02f * frustumSize);
            float gb = -(dot3(Nv0, Xv0) * ga);
            float gradient = g0, sum = 1.0f;
            for (int i = -2; i <= 2; i++)
                for (int j = -2; j <= 2; j++) {
                    if (i == 0 && j == 0)
                        continue;
                    int px = x + i * step, py = y + j * step;
                    float u = ((float)px + 0.5f) * d.inv_size[0], v = ((float)py + 0.5f) * d.inv_size[1];
                    int cx = px < 0 ? 0 : (px >= W ? W - 1 : px), cy = py < 0 ? 0 : (py >= H ? H - 1 : py); // gNearestClamp
                    const H4* t = texel(in, d.pitch_bytes, cx, cy);
                    float w = gauss_weight(i * i + j * j);
                    float z = f16_to_f32(t->v[3]) / FP16_VIEWZ_SCALE;
                    f3 Xv = reconstruct_view(d.camera_frustum, u, v, z, d.ortho_mode);
                    float NoX = dot3(Nv0, Xv);
                    w *= smoothstep01(1.0f - absf(fma_(NoX, ga, gb)));
                    f3 Nv = oct_decode({f16_to_f32(t->v[1]), f16_to_f32(t->v[2])});
                    float NoN = sat(dot3(Nv0, Nv));
                    w *= NoN * NoN;
                    gradient = fma_(f16_to_f32(t->v[0]), w, gradient);
                    sum += w;
                }
            gradient /= sum;
            if (last) { // :86-103
                gradient = hdr_to_linear_uncharted(gradient);
                gradient = 1.0f - to_srgb(sat(gradient));
                if (d.relax)
                    gradient *= gradient;
                float dither = bayer4x4((uint32_t)x, (uint32_t)y, d.frame_index);
                gradient += (dither - 0.5f) / (float)d.max_accumulated_frame_num;
            }
            o.v[0] = to_h(sat(gradient));
            *texel(out, d.pitch_bytes, x, y) = o;
        }
}

inline float sh_resolve_scale(float c0, f3 c1, f3 dir) {
    float Y = fmax2(fma_(0.5f, c0, dot3(dir, c1)), 0.0f) * (2.0f / 3.0f);
    return Y / fmax2(c0, 1e-6f);
}
inline void store4(void* base, uint32_t pitch, int x, int y, float a, float b, float c, float d) {
    H4* o = texel(base, pitch, x, y);
    o->v[0] = to_h(a);
    o->v[1] = to_h(b);
    o->v[2] = to_h(c);
    o->v[3] = to_h(d);
}

} // namespace

extern "C" {

__attribute__((visibility("default"))) int orc_confidence_blur(const nrdhip_confidence_blur_desc* d, void* /*stream*/) {
    if (!d || !d->ping || !d->pong || !d->width || !d->height || d->pitch_bytes < (uint32_t)d->width * 8u || d->first_pass + d->passes_num > 5u)
        return 2;
    for (uint32_t i = d->first_pass; i < d->first_pass + d->passes_num; i++) {
        bool even = (i & 1u) == 0u; // Source/NRDSample.cpp:4004-4008
        confidence_pass(*d, even ? d->ping : d->pong, even ? d->pong : d->ping, (int)(1u + i));
    }
    return 0;
}

__attribute__((visibility("default"))) int orc_backend_unpack(const nrdhip_unpack_desc* d, void* /*stream*/) {
    if (!d || !d->width || !d->height || d->mode > NRDHIP_UNPACK_SH)
        return 2;
    if (d->mode == NRDHIP_UNPACK_SH && ((d->diff && d->out_diff && !d->diff_sh1 && d->resolve) || (d->spec && d->out_spec && !d->spec_sh1 && d->resolve) ||
                                        (d->resolve && !d->normal_roughness)))
        return 2;
    if (d->out_shadow && (!d->shadow || (d->shadow_bytes_per_texel != 1 && d->shadow_bytes_per_texel != 4)))
        return 2;
    const bool sh = d->mode == NRDHIP_UNPACK_SH;
    for (int y = 0; y < d->height; y++)
        for (int x = 0; x < d->width; x++) {
            if (d->out_shadow) { // SIGMA_BackEnd_UnpackShadow (Composition.cs.hlsl:57-64)
                float s[4];
                const uint8_t* sp = (const uint8_t*)d->shadow + (size_t)y * d->shadow_pitch + (size_t)x * d->shadow_bytes_per_texel;
                for (int k = 0; k < 4; k++)
                    s[k] = (float)sp[d->shadow_bytes_per_texel == 1 ? 0 : k] * (1.0f / 255.0f);
                store4(d->out_shadow, d->out_shadow_pitch, x, y, s[0] * s[0], s[1] * s[1], s[2] * s[2], s[3] * s[3]);
            }
            f3 N = {0.0f, 0.0f, 1.0f}, V = {0.0f, 0.0f, 1.0f};
            float roughness = 1.0f;
            if (sh && d->resolve) {
                uint32_t nr = *(const uint32_t*)((const uint8_t*)d->normal_roughness + (size_t)y * d->normal_roughness_pitch + (size_t)x * 4);
                N = oct_decode({(float)(nr & 1023u) * (1.0f / 1023.0f), (float)((nr >> 10) & 1023u) * (1.0f / 1023.0f)});
                roughness = (float)((nr >> 20) & 1023u) * (1.0f / 1023.0f);
                float u = ((float)x + 0.5f) * d->inv_rect_size[0], v = ((float)y + 0.5f) * d->inv_rect_size[1];
                f3 Xv = {fma_(u, d->camera_frustum[2], d->camera_frustum[0]), fma_(v, d->camera_frustum[3], d->camera_frustum[1]), 1.0f};
                V = normalize3(rot3(d->view_to_world, mul3(Xv, -1.0f)));
            }
            for (int sig = 0; sig < 2; sig++) {
                const void* in = sig ? d->spec : d->diff;
                uint32_t inPitch = sig ? d->spec_pitch : d->diff_pitch;
                const void* in1 = sig ? d->spec_sh1 : d->diff_sh1;
                uint32_t in1Pitch = sig ? d->spec_sh1_pitch : d->diff_sh1_pitch;
                void* out = sig ? d->out_spec : d->out_diff;
                uint32_t outPitch = sig ? d->out_spec_pitch : d->out_diff_pitch;
                if (!in || !out)
                    continue;
                if (d->mode == NRDHIP_UNPACK_OCCLUSION) { // Composition.cs.hlsl:124-126
                    float h = (float)*(const uint16_t*)((const uint8_t*)in + (size_t)y * inPitch + (size_t)x * 2) * (1.0f / 65535.0f);
                    store4(out, outPitch, x, y, h, h, h, h);
                    continue;
                }
                const H4* t = texel(in, inPitch, x, y);
                float vx = f16_to_f32(t->v[0]), vy = f16_to_f32(t->v[1]), vz = f16_to_f32(t->v[2]), vw = f16_to_f32(t->v[3]);
                f3 rgb = d->relax ? f3{vx, vy, vz} : ycocg_to_linear({vx, vy, vz});
                if (!d->relax)
                    rgb = {fmax2(rgb.x, 0.0f), fmax2(rgb.y, 0.0f), fmax2(rgb.z, 0.0f)};
                if (sh && d->resolve) { // :85-122 without the re-jitter step
                    const H4* t1 = texel(in1, in1Pitch, x, y);
                    f3 c1 = {f16_to_f32(t1->v[0]), f16_to_f32(t1->v[1]), f16_to_f32(t1->v[2])};
                    f3 dir = N;
                    if (sig) {
                        float NoV = dot3(N, V);
                        f3 R = sub3(mul3(N, 2.0f * NoV), V);
                        dir = normalize3(add3(N, mul3(sub3(R, N), spec_dominant_factor(roughness))));
                    }
                    float Y = d->relax ? fma_(0.25f, vx, fma_(0.5f, vy, 0.25f * vz)) : vx;
                    rgb = mul3(rgb, sh_resolve_scale(Y, c1, dir));
                }
                store4(out, outPitch, x, y, rgb.x, rgb.y, rgb.z, d->relax ? 0.318309886f : vw);
            }
        }
    return 0;
}
}
