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

// ---- TAA (Shaders/Taa.cs.hlsl:11-159; ApplyTonemap / BicubicFilterNoCorners: Shaders/Shared.hlsli:337-387; Color::ClampAabb,
// Color::RgbToXyz, Math::PositiveRcp: MathLib [RECOLLECTION]) ----------------------------------------------------------------
struct Taa {
    const nrdhip_taa_desc& d;
    int W, H, RW, RH;
    float Wp, Hp, invW, invH, invRW, invRH;
};
inline float positive_rcp(float x) { return rcp_(fmax2(x, 1e-15f)); }
inline f4 ld4(const void* base, uint32_t pitch, int x, int y) {
    const H4* t = texel(base, pitch, x, y);
    return {f16_to_f32(t->v[0]), f16_to_f32(t->v[1]), f16_to_f32(t->v[2]), f16_to_f32(t->v[3])};
}
inline f4 mul4(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline f4 fma4(f4 a, float s, f4 c) { return {fma_(a.x, s, c.x), fma_(a.y, s, c.y), fma_(a.z, s, c.z), fma_(a.w, s, c.w)}; }
inline f4 lerp4(f4 a, f4 b, float t) { return {lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t)}; }
inline f3 taa_tonemap(const Taa& k, f3 c) {
    if (!k.d.tonemap)
        return c;
    float s = k.d.hdr_scale;
    return {s * hdr_to_linear_uncharted(c.x), s * hdr_to_linear_uncharted(c.y), s * hdr_to_linear_uncharted(c.z)};
}
inline f4 sample_linear_clamp(const Taa& k, float u, float v) {
    const int w = k.RW, h = k.RH;
    float x = fma_(u, (float)w, -0.5f), y = fma_(v, (float)h, -0.5f);
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    x0 = clampf(x0, -1.0f, (float)w);
    y0 = clampf(y0, -1.0f, (float)h);
    auto cl = [](int a, int n) { return a < 0 ? 0 : (a > n - 1 ? n - 1 : a); };
    int ix0 = cl((int)x0, w), ix1 = cl((int)x0 + 1, w), iy0 = cl((int)y0, h), iy1 = cl((int)y0 + 1, h);
    f4 a = ld4(k.d.history, k.d.history_pitch, ix0, iy0), b = ld4(k.d.history, k.d.history_pitch, ix1, iy0);
    f4 c = ld4(k.d.history, k.d.history_pitch, ix0, iy1), e = ld4(k.d.history, k.d.history_pitch, ix1, iy1);
    return lerp4(lerp4(a, b, fx), lerp4(c, e, fx), fy);
}
f4 bicubic_no_corners(const Taa& k, float sx, float sy) {
    const float sh = 0.66f; // TAA_HISTORY_SHARPNESS
    float cx = floorf(sx - 0.5f) + 0.5f, cy = floorf(sy - 0.5f) + 0.5f;
    float f[2] = {sat(sx - cx), sat(sy - cy)};
    float w0[2], w3[2], wl2[2], tc2[2];
    const float c[2] = {cx, cy}, inv[2] = {k.invRW, k.invRH};
    for (int i = 0; i < 2; i++) {
        float f1 = f[i], f2 = f1 * f1, f3v = f1 * f2;
        w0[i] = -sh * f3v + 2.0f * sh * f2 - sh * f1;
        float w1 = (2.0f - sh) * f3v - (3.0f - sh) * f2 + 1.0f;
        float w2 = -(2.0f - sh) * f3v + (3.0f - 2.0f * sh) * f2 + sh * f1;
        w3[i] = sh * f3v - sh * f2;
        wl2[i] = w1 + w2;
        tc2[i] = inv[i] * (c[i] + w2 * positive_rcp(wl2[i]));
    }
    float tc0x = k.invRW * (cx - 1.0f), tc0y = k.invRH * (cy - 1.0f), tc3x = k.invRW * (cx + 2.0f), tc3y = k.invRH * (cy + 2.0f);
    float w = wl2[0] * w0[1];
    f4 color = mul4(sample_linear_clamp(k, tc2[0], tc0y), w);
    float sum = w;
    w = w0[0] * wl2[1];
    color = fma4(sample_linear_clamp(k, tc0x, tc2[1]), w, color);
    sum += w;
    w = wl2[0] * wl2[1];
    color = fma4(sample_linear_clamp(k, tc2[0], tc2[1]), w, color);
    sum += w;
    w = w3[0] * wl2[1];
    color = fma4(sample_linear_clamp(k, tc3x, tc2[1]), w, color);
    sum += w;
    w = wl2[0] * w3[1];
    color = fma4(sample_linear_clamp(k, tc2[0], tc3y), w, color);
    sum += w;
    return mul4(color, positive_rcp(sum));
}
inline f3 rgb_to_xyz(f3 c) {
    return {100.0f * fma_(0.1804808f, c.z, fma_(0.3575843f, c.y, 0.4123908f * c.x)), 100.0f * fma_(0.0721923f, c.z, fma_(0.7151687f, c.y, 0.2126390f * c.x)),
            100.0f * fma_(0.9505322f, c.z, fma_(0.1191948f, c.y, 0.0193308f * c.x))};
}
inline f3 xyz_to_lab(f3 x) { // Taa.cs.hlsl:43-54 ("l" is computed from the already transformed y, as in the shader); pow(x, 0.333333) = cbrt_pos_ (ledger row 19)
    x = {x.x * (1.0f / 95.0489f), x.y * (1.0f / 100.0f), x.z * (1.0f / 108.8840f)};
    float fx = x.x > 0.008856f ? cbrt_pos_(x.x) : fma_(7.787f, x.x, 16.0f / 116.0f);
    float fy = x.y > 0.008856f ? cbrt_pos_(x.y) : fma_(7.787f, x.y, 16.0f / 116.0f);
    float fz = x.z > 0.008856f ? cbrt_pos_(x.z) : fma_(7.787f, x.z, 16.0f / 116.0f);
    float l = fy > 0.008856f ? fma_(116.0f, cbrt_pos_(fy), -16.0f) : 903.3f * fy;
    return {l, 500.0f * (fx - fy), 200.0f * (fy - fz)};
}
inline f3 clamp_aabb(f3 center, f3 ext, f3 prev) {
    f3 d = sub3(prev, center);
    f3 dn = {absf(d.x * positive_rcp(ext.x)), absf(d.y * positive_rcp(ext.y)), absf(d.z * positive_rcp(ext.z))};
    float maxd = fmax2(dn.x, fmax2(dn.y, dn.z));
    float r = positive_rcp(maxd);
    f3 t = {fma_(d.x, r, center.x), fma_(d.y, r, center.y), fma_(d.z, r, center.z)};
    return maxd > 1.0f ? t : prev;
}

} // namespace

extern "C" {

__attribute__((visibility("default"))) int orc_taa(const nrdhip_taa_desc* dp, void* /*stream*/) {
    if (!dp || !dp->mv || !dp->composed || !dp->history || !dp->result || !dp->rect_width || !dp->rect_height || !dp->render_width || !dp->render_height ||
        dp->rect_width > dp->render_width || dp->rect_height > dp->render_height)
        return 2;
    const nrdhip_taa_desc& d = *dp;
    Taa k{d, d.rect_width, d.rect_height, d.render_width, d.render_height, 0, 0, 0, 0, 0, 0};
    k.Wp = (float)(d.rect_width_prev ? d.rect_width_prev : d.rect_width);
    k.Hp = (float)(d.rect_height_prev ? d.rect_height_prev : d.rect_height);
    k.invW = 1.0f / (float)k.W;
    k.invH = 1.0f / (float)k.H;
    k.invRW = 1.0f / (float)k.RW;
    k.invRH = 1.0f / (float)k.RH;
    auto cl = [](int a, int n) { return a < 0 ? 0 : (a > n - 1 ? n - 1 : a); };
    for (int y = 0; y < k.H; y++)
        for (int x = 0; x < k.W; x++) {
            float u = ((float)x + 0.5f) * k.invW, v = ((float)y + 0.5f) * k.invH;
            if (u > 1.0f || v > 1.0f)
                continue;
            // the 20x20 shared-memory tile of the shader, read straight from the (clamped) planes
            auto color_at = [&](int px, int py) {
                f4 c = ld4(d.composed, d.composed_pitch, cl(px, k.W), cl(py, k.H));
                return taa_tonemap(k, {c.x, c.y, c.z});
            };
            auto mv_at = [&](int px, int py) {
                f4 m = ld4(d.mv, d.mv_pitch, cl(px, k.W), cl(py, k.H));
                return f3{m.x, m.y, m.w};
            };
            float sum = 0.0f;
            f3 m1 = {0, 0, 0}, m2 = {0, 0, 0}, input = {0, 0, 0};
            float centerZ = mv_at(x, y).z;
            float minViewZ = absf(centerZ);
            int offx = 2, offy = 2;
            const bool want5x5 = centerZ < 0.0f;
            for (int dy = 0; dy <= 4; dy++)
                for (int dx = 0; dx <= 4; dx++) {
                    const bool border = dx == 0 || dx == 4 || dy == 0 || dy == 4;
                    if (border && !want5x5)
                        continue;
                    f3 c = color_at(x + dx - 2, y + dy - 2);
                    float viewZ = absf(mv_at(x + dx - 2, y + dy - 2).z);
                    if (dx == 2 && dy == 2)
                        input = c;
                    else if (viewZ < minViewZ) {
                        minViewZ = viewZ;
                        offx = dx;
                        offy = dy;
                    }
                    const int qx = dx / 2 - 1, qy = dy / 2 - 1; // integer division of the shader (:103)
                    const int r2 = qx * qx + qy * qy;
                    const float w = r2 == 0 ? 1.0f : (r2 == 1 ? 0.36787944f : 0.13533528f);
                    m1 = {fma_(c.x, w, m1.x), fma_(c.y, w, m1.y), fma_(c.z, w, m1.z)};
                    m2 = {fma_(c.x * c.x, w, m2.x), fma_(c.y * c.y, w, m2.y), fma_(c.z * c.z, w, m2.z)};
                    sum += w;
                }
            float rs = rcp_(sum);
            m1 = mul3(m1, rs);
            m2 = mul3(m2, rs);
            f3 sigma = {sqrt_(absf(m2.x - m1.x * m1.x)) * 2.0f, sqrt_(absf(m2.y - m1.y * m1.y)) * 2.0f, sqrt_(absf(m2.z - m1.z * m1.z)) * 2.0f};
            f3 mvn = mv_at(x + offx - 2, y + offy - 2);
            float pu = fma_(mvn.x, k.invW, u), pv = fma_(mvn.y, k.invH, v);
            f4 history = bicubic_no_corners(k, sat(pu) * k.Wp, sat(pv) * k.Hp);
            f3 hist = {fmax2(history.x, 0.0f), fmax2(history.y, 0.0f), fmax2(history.z, 0.0f)};
            float mixRate = sat(history.w);
            mixRate = mixRate * rcp_(1.0f + mixRate);
            bool inScreen = sat(pu) == pu && sat(pv) == pv;
            mixRate = inScreen ? mixRate : 1.0f;
            f3 clamped = clamp_aabb(m1, sigma, hist);
            f3 a = xyz_to_lab(rgb_to_xyz(clamped)), b = xyz_to_lab(rgb_to_xyz(hist));
            f3 dl = sub3(a, b);
            float diff = sqrt_(dot3(dl, dl)) * (1.0f / (2.3f * 3.0f));
            mixRate = sat(mixRate + diff);
            float t = fmax2(mixRate, d.taa);
            store4(d.result, d.result_pitch, x, y, lerpf(clamped.x, input.x, t), lerpf(clamped.y, input.y, t), lerpf(clamped.z, input.z, t), mixRate);
        }
    return 0;
}


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
