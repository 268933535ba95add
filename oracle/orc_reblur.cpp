// orc_reblur.cpp - CPU ORACLE of REBLUR_DIFFUSE / REBLUR_SPECULAR / REBLUR_DIFFUSE_SPECULAR.
// Test infrastructure only; PARITY UNPINNED vs upstream NRD (External/NRD is absent from the reference tree,
// SURVEY.md section 0). This file is the frozen restatement the HIP kernels are checked against.
//
// Contract followed (reference call sites, relative to /root/reference):
//   slots & formats ............ Source/NRDSample.cpp:447-462 (bind), :2971-2990 (formats)
//   settings fed ............... Source/NRDSample.cpp:563-585, :2169-2184, :4090-4126
//   CommonSettings ............. Source/NRDSample.cpp:3835-3876
//   input encodings ............ Shaders/TraceOpaque.cs.hlsl:609-620 (MV, viewZ, sky = +-INF), :657 (normal/roughness/material),
//                                :418-421 + :756-757 (YCoCg radiance + normalised hit distance), Shaders/Shared.hlsli:318-335 (2.5D motion)
//   output decoding ............ Shaders/Composition.cs.hlsl:165-166
// Pass graph: ClassifyTiles(+guide packing) -> PrePass -> TemporalAccumulation -> HistoryFix ->
// Blur -> PostBlur -> TemporalStabilization (+ split screen). DESIGN.md section "REBLUR" documents every formula.
#include "orc_core.h"
#include "../include/nrdhip.h"

namespace orc {

namespace {

// pool plane indices relative to permBase / transBase
enum Perm { P_GUIDE_A, P_GUIDE_B, P_DATA1_A, P_DATA1_B, P_HIST, P_FAST_A, P_FAST_B, P_STAB_A, P_STAB_B, P_NUM };
enum Trans { T_TILES, T_TMP1, T_TMP2, T_DATA1, T_DATA2, T_HITTRACK, T_PREP_D, T_PREP_S, T_PREP_D1, T_PREP_S1, T_NUM,
             // REBLUR only (RELAX keeps its A-trous ping-pong at these indices): tap texels of Blur / PostBlur, see TapTexel below.
             // _A: HistoryFix -> Blur, _B: Blur -> PostBlur; one plane per signal
             T_TAP_D_A = T_NUM, T_TAP_S_A, T_TAP_D_B, T_TAP_S_B };

const float MAX_ACCUM = 63.0f;
const float MIN_CONVERGED_RADIUS_SCALE = 0.25f;
const float POST_BLUR_RADIUS_SCALE = 2.0f;
const float NORMAL_ANGLE_MIN = 0.02f;
const float PREV_NORMAL_COS = 0.7f;

static inline f4 add4(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline f4 mul4(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline f4 fma4(f4 a, float s, f4 c) { return {fma_(a.x, s, c.x), fma_(a.y, s, c.y), fma_(a.z, s, c.z), fma_(a.w, s, c.w)}; }
static inline f4 lerp4(f4 a, f4 b, float t) { return {lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t)}; }
// OCCLUSION variants: the signal is the normalised hit distance alone (R16_UNORM or R16F plane); internally it travels as
// {h, 0, 0, h} so every luma-based stage works on it unchanged (Source/NRDSample.cpp:488-501 binds IN/OUT_*_HITDIST)
static inline f4 load_signal(const Plane& P, int x, int y, int off, bool occlusion) {
    if (!occlusion)
        return ld_h4(P, x, y, off);
    float h = P.fmt == (uint32_t)nrd::Format::R16_SFLOAT ? ld_h(P, x, y) : (float)ld_u16(P, x, y) * (1.0f / 65535.0f);
    return {h, 0.0f, 0.0f, h};
}
static inline void store_signal(const Plane& P, int x, int y, f4 v, bool occlusion) {
    if (!occlusion) {
        st_h4(P, x, y, v);
        return;
    }
    if (P.fmt == (uint32_t)nrd::Format::R16_SFLOAT)
        st_h(P, x, y, v.x);
    else
        st_u16(P, x, y, (uint16_t)floorf(fma_(sat(v.x), 65535.0f, 0.5f)));
}
// DIRECTIONAL_OCCLUSION split-screen passthrough: the noisy {direction * h, h} texel rebuilt from its prepared SH0 / SH1 halves
static inline f4 dir_pass(const Plane& sh0, const Plane& sh1, int x, int y) {
    f4 a = ld_h4(sh0, x, y), b = ld_h4(sh1, x, y);
    return {b.x, b.y, b.z, a.x};
}
// OUT_DIFF_DIRECTION_HITDIST texel in the bound format
static inline void st_dir(const Plane& P, int x, int y, f4 v) {
    if (P.fmt == (uint32_t)nrd::Format::RGBA16_SNORM)
        st_sn4(P, x, y, v);
    else
        st_h4(P, x, y, v);
}
static inline nrd::ResourceType in_slot(const DenoiserState& d, bool spec) {
    using RT = nrd::ResourceType;
    if (d.dirOcc)
        return RT::IN_DIFF_DIRECTION_HITDIST;
    if (d.sh)
        return spec ? RT::IN_SPEC_SH0 : RT::IN_DIFF_SH0;
    return d.occlusion ? (spec ? RT::IN_SPEC_HITDIST : RT::IN_DIFF_HITDIST) : (spec ? RT::IN_SPEC_RADIANCE_HITDIST : RT::IN_DIFF_RADIANCE_HITDIST);
}
static inline nrd::ResourceType in1_slot(bool spec) { return spec ? nrd::ResourceType::IN_SPEC_SH1 : nrd::ResourceType::IN_DIFF_SH1; }
static inline nrd::ResourceType out1_slot(bool spec) { return spec ? nrd::ResourceType::OUT_SPEC_SH1 : nrd::ResourceType::OUT_DIFF_SH1; }
static inline nrd::ResourceType out_slot(const DenoiserState& d, bool spec) {
    using RT = nrd::ResourceType;
    if (d.dirOcc)
        return RT::OUT_DIFF_DIRECTION_HITDIST;
    if (d.sh)
        return spec ? RT::OUT_SPEC_SH0 : RT::OUT_DIFF_SH0;
    return d.occlusion ? (spec ? RT::OUT_SPEC_HITDIST : RT::OUT_DIFF_HITDIST) : (spec ? RT::OUT_SPEC_RADIANCE_HITDIST : RT::OUT_DIFF_RADIANCE_HITDIST);
}
static inline f4 rgb_to_ycocg4(f4 v) {
    f3 c = linear_to_ycocg({v.x, v.y, v.z});
    return {c.x, c.y, c.z, v.w};
}

static inline void unpack_data1(uint16_t v, float& diffA, float& specA) {
    diffA = (float)(v & 0xffu) * 0.25f;
    specA = (float)(v >> 8) * 0.25f;
}
static inline uint16_t pack_data1(float diffA, float specA) {
    uint32_t d = (uint32_t)floorf(fma_(clampf(diffA, 0.0f, MAX_ACCUM), 4.0f, 0.5f));
    uint32_t s = (uint32_t)floorf(fma_(clampf(specA, 0.0f, MAX_ACCUM), 4.0f, 0.5f));
    return (uint16_t)(d | (s << 8));
}

struct Ctx {
    Instance& I;
    DenoiserState& d;
    const Consts& c;
    int cur; // ping-pong parity
    const Plane& perm(int i) const { return I.perm[d.permBase + i]; }
    const Plane& trans(int i) const { return I.trans[d.transBase + i]; }
    const Plane& slot(nrd::ResourceType t) const { return I.slots[(size_t)t]; }
    const Plane& guide() const { return perm(P_GUIDE_A + cur); }
    const Plane& guidePrev() const { return perm(P_GUIDE_A + (cur ^ 1)); }
    int sigDiff() const { return 0; }
    int sigSpec() const { return d.hasDiff ? 1 : 0; }
};

// --------------------------------------------------------------------------------------------------
// K0 ClassifyTiles + guide packing (orc_core.h encode_guide); tile = 1 if every pixel of the tile is sky
// --------------------------------------------------------------------------------------------------
void classify_tiles(Instance& I, DenoiserState& d, const Consts& c, int ty0, int ty1) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const int sb = d.sh ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    (void)sb;
    const Plane& inZ = k.slot(nrd::ResourceType::IN_VIEWZ);
    const Plane& inNR = k.slot(nrd::ResourceType::IN_NORMAL_ROUGHNESS);
    const Plane& G = k.guide();
    const Plane& T = k.trans(T_TILES);
    float zs = I.common.viewZScale;
    int tilesX = (c.W + 15) / 16;
    for (int ty = ty0; ty < ty1; ty++)
        for (int tx = 0; tx < tilesX; tx++) {
            bool allSky = true;
            for (int j = 0; j < 16; j++)
                for (int i = 0; i < 16; i++) {
                    int x = tx * 16 + i, y = ty * 16 + j;
                    if (x >= c.W || y >= c.resH || y + c.yOff >= c.H || y + c.yOff < 0)
                        continue;
                    float z = ld_f32(inZ, x, y) * zs;
                    if (store_guide(G, x, y, z, ld_u32(inNR, x, y), c.denoisingRange))
                        allSky = false;
                }
            *texel(T, tx, ty) = allSky ? 1 : 0;
        }
}

// per-pixel geometry shared by the bilateral passes
struct PixelGeo {
    f3 Xv, Nv;
    float rx, ry; // perspective: the pixel's view ray (rx, ry, 1); orthographic: its view-space xy
    float absZ, frustumSize;
    float ga0, gax, gay, geoB; // plane-distance weight: |zs * (ga0 + gax px + gay gy) + geoB|
    bool ortho;                // orthographic: |zs * geoB + (ga0 + gax px + gay gy)| (geoB = z coefficient, ga0 absorbs the offset)
};

static inline PixelGeo pixel_geo(const Consts& c, const Guide& g, int x, int gy, float planeDistSensitivity) {
    PixelGeo p;
    p.rx = fma_(c.pv[2], (float)x, c.pv[0]);
    p.ry = fma_(c.pv[3], (float)gy, c.pv[1]);
    p.Xv = {(c.ortho ? 1.0f : g.z) * p.rx, (c.ortho ? 1.0f : g.z) * p.ry, g.z}; // == reconstruct_px(c.pv, x, gy, g.z)
    p.Nv = rot3(c.w2v, g.n);
    p.absZ = absf(g.z);
    p.ortho = c.ortho;
    p.frustumSize = c.minRectDimMulUnproject * (c.ortho ? 1.0f : p.absZ);
    float geoA = wrcp_(planeDistSensitivity * p.frustumSize);
    p.gax = p.Nv.x * c.pv[2] * geoA;
    p.gay = p.Nv.y * c.pv[3] * geoA;
    if (c.ortho) {
        p.ga0 = (fma_(p.Nv.x, c.pv[0], p.Nv.y * c.pv[1]) - dot3(p.Nv, p.Xv)) * geoA;
        p.geoB = p.Nv.z * geoA;
    } else {
        p.ga0 = fma_(p.Nv.x, c.pv[0], fma_(p.Nv.y, c.pv[1], p.Nv.z)) * geoA;
        p.geoB = -dot3(p.Nv, p.Xv) * geoA;
    }
    return p;
}
static inline float geo_weight(const PixelGeo& p, float px, float gy, float zs) {
    float ga = fma_(p.gax, px, fma_(p.gay, gy, p.ga0));
    return smoothstep01(1.0f - absf(p.ortho ? fma_(zs, p.geoB, ga) : fma_(zs, ga, p.geoB)));
}
// Hair (CommonSettings::strandMaterialID / strandThickness, Source/NRDSample.cpp:3871-3872; the sample's own guide treatment:
// Shaders/TraceOpaque.cs.hlsl:644-649): per-pixel normals of strands thinner than a pixel are unreliable, so the normal-weight
// parameter of such pixels is scaled by lerp(0.25, 1, saturate(strandThickness / pixel world size))
static inline float strand_normal_relax(const Consts& c, uint32_t mat, float absZ) {
    if (mat != c.strandMat)
        return 1.0f;
    return lerpf(0.25f, 1.0f, sat(c.strandThickness * wrcp_(c.unproject * (c.ortho ? 1.0f : absZ))));
}
// unit vector from the view-space point toward the viewer
static inline f3 to_viewer(const Consts& c, f3 Xv) {
    if (c.ortho)
        return {0.0f, 0.0f, Xv.z >= 0.0f ? -1.0f : 1.0f};
    return mul3(normalize3(Xv), -1.0f);
}

// --------------------------------------------------------------------------------------------------
// Spatial filter shared by PrePass / Blur / PostBlur
// --------------------------------------------------------------------------------------------------
enum Variant { PRE = 0, BLUR = 1, POST = 2 };

// Kernel set-up of a signal: the part that hangs on the pixel's geometry and roughness only, not on the pass (csrc/nrd_reblur.hip KernelUnit)
struct KernelUnit {
    float j[4];                           // pixel offsets of the kernel's tangent / bitangent per pixel of blur radius
    float smc, angle0, roughA, hitFactor; // GetSpecMagicCurve(roughness), lobe half angle, 1 / roughness tolerance, hit distance factor
};
static inline KernelUnit kernel_unit(const Consts& c, const nrd::ReblurSettings& s, const PixelGeo& pg, float z, f3 V, float rough, bool isSpec) {
    const float* hp = &s.hitDistanceParameters.A;
    KernelUnit k;
    f3 T, B;
    basis3(pg.Nv, T, B);
    if (isSpec) {
        const float NoV = dot3(pg.Nv, V);
        const float df = spec_dominant_factor(rough);
        // dominant direction D = normalize(N + (R - N) df), R = 2 NoV N - V the mirror direction: N (1 + (2 NoV - 1) df) - V df
        const float alpha = fma_(fma_(NoV, 2.0f, -1.0f), df, 1.0f);
        const f3 D = normalize3({fma_(pg.Nv.x, alpha, -(V.x * df)), fma_(pg.Nv.y, alpha, -(V.y * df)), fma_(pg.Nv.z, alpha, -(V.z * df))});
        const float NoD = dot3(pg.Nv, D);
        if (NoD < 0.999f && rough < 0.95f) {
            const float n2 = 2.0f * NoD;
            const f3 Dr = {fma_(pg.Nv.x, n2, -D.x), fma_(pg.Nv.y, n2, -D.y), fma_(pg.Nv.z, n2, -D.z)}; // D mirrored at N
            T = normalize3(cross3(D, pg.Nv)); // == cross(N, Dr)
            B = cross3(Dr, T);
            T = mul3(T, lerpf(fma_(rough, 0.5f, 0.5f), 1.0f, NoD));
        }
        k.smc = spec_magic_curve(rough);
        k.angle0 = spec_lobe_half_angle(rough);
        k.roughA = wrcp_(lerpf(0.01f, 1.0f, sat(rough * s.roughnessFraction)));
        k.hitFactor = reblur_hitdist_factor(hp, rough);
    } else {
        k.smc = 1.0f;
        k.angle0 = spec_lobe_half_angle(1.0f);
        k.roughA = 0.0f;
        k.hitFactor = reblur_hitdist_factor(hp, 1.0f);
    }
    kernel_basis_px(c, z, pg.rx, pg.ry, T, B, k.j);
    return k;
}

// --------------------------------------------------------------------------------------------------
// K1 PrepareInputs - recorded only when checkerboardMode != OFF or hitDistanceReconstructionMode != OFF (the sample's default
// operating mode: tracingMode RESOLUTION_HALF -> CheckerboardMode::WHITE, Source/NRDSample.cpp:267, :545-548, :565-568).
// Produces dense, full-resolution RGBA16F copies of the noisy inputs for the PrePass:
//  * checkerboard resolve: the half-width inputs hold pixel (x, y) at texel (x >> 1, y) on the squares of
//    Sequence::CheckerBoard(pixelPos, frameIndex) that carry the signal (WHITE: diffuse on 1, specular on 0;
//    Shaders/TraceOpaque.cs.hlsl:482-508); a pixel of the other colour takes the depth-weighted mean of its left / right
//    neighbours;
//  * hit distance reconstruction (AREA_3X3 / AREA_5X5): a texel without hit distance (w == 0) takes the bilateral mean
//    (plane distance x normal [x roughness]) of the valid hit distances around it.
// --------------------------------------------------------------------------------------------------
struct PrepareMode {
    bool any, checker;
    int phase[2]; // per signal (0 diffuse, 1 specular): checkerboard value carrying it, 2 = every pixel
    int radius;   // hit distance reconstruction radius, 0 = off
    bool sh1;     // the SH1 texels are (re)written too: checkerboarded SH inputs, or DIRECTIONAL_OCCLUSION (its single
                  // {direction * h, h} input texel is always split into SH0 = {h,0,0,h} and SH1 = {direction * h, 0} here)
};
static inline PrepareMode prepare_mode(const DenoiserState& d) {
    const nrd::ReblurSettings& s = d.reblur;
    PrepareMode m;
    m.checker = s.checkerboardMode != nrd::CheckerboardMode::OFF;
    bool white = s.checkerboardMode == nrd::CheckerboardMode::WHITE;
    m.phase[0] = !m.checker ? 2 : (white ? 1 : 0);
    m.phase[1] = !m.checker ? 2 : (white ? 0 : 1);
    m.radius = s.hitDistanceReconstructionMode == nrd::HitDistanceReconstructionMode::OFF ? 0 : (s.hitDistanceReconstructionMode == nrd::HitDistanceReconstructionMode::AREA_3X3 ? 1 : 2);
    m.any = m.checker || m.radius > 0 || d.dirOcc;
    m.sh1 = d.sh && (m.checker || d.dirOcc);
    return m;
}
static inline bool has_data(int phase, int x, int gy, uint32_t frameIndex) { return phase == 2 || ((((uint32_t)x ^ (uint32_t)gy) ^ frameIndex) & 1u) == (uint32_t)phase; }

void prepare_inputs(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const nrd::ReblurSettings& s = d.reblur;
    const PrepareMode m = prepare_mode(d);
    const Plane& G = k.guide();
    const bool sh1 = m.sh1;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c.W; x++) {
            Guide g = load_guide(G, x, y, c.denoisingRange);
            const int gy0 = y + c.yOff;
            for (int sig = 0; sig < d.nsig; sig++) {
                const bool isSpec = (sig == k.sigSpec()) && d.hasSpec;
                const int si = isSpec ? 1 : 0;
                const Plane& in = k.slot(in_slot(d, isSpec));
                const Plane& out = k.trans(T_PREP_D + si);
                const Plane* in1 = (sh1 && !d.dirOcc) ? &k.slot(in1_slot(isSpec)) : nullptr;
                // one input position -> (SH0-like signal, SH1 texel)
                auto load_pair = [&](int sx, int sy, f4& a, f4& b) {
                    if (d.dirOcc) { // {direction * h, h}
                        f4 t = in.fmt == (uint32_t)nrd::Format::RGBA16_SNORM ? ld_sn4(in, sx, sy) : ld_h4(in, sx, sy, 0);
                        a = {t.w, 0.0f, 0.0f, t.w};
                        b = {t.x, t.y, t.z, 0.0f};
                    } else {
                        a = load_signal(in, sx, sy, 0, d.occlusion);
                        b = sh1 ? ld_h4(*in1, sx, sy, 0) : f4{0, 0, 0, 0};
                    }
                };
                const Plane* out1 = sh1 ? &k.trans(T_PREP_D1 + si) : nullptr;
                if (g.sky) {
                    st_h4(out, x, y, {0, 0, 0, 0});
                    if (sh1)
                        st_h4(*out1, x, y, {0, 0, 0, 0});
                    continue;
                }
                const int phase = m.phase[si];
                f4 v = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
                if (has_data(phase, x, gy0, c.frameIndex)) {
                    int sx = m.checker ? x >> 1 : x;
                    load_pair(sx, y, v, v1);
                } else { // checkerboard resolve from the left / right neighbours (they carry this signal)
                    float invDz = rcp_(0.03f * fmax2(absf(g.z), 1e-6f));
                    float wn[2];
                    bool ok[2];
                    f4 vn[2], v1n[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                    for (int n = 0; n < 2; n++) {
                        int px = x + (n ? 1 : -1);
                        ok[n] = px >= 0 && px < c.W;
                        int cpx = px < 0 ? 0 : (px >= c.W ? c.W - 1 : px);
                        Guide gn = load_guide(G, cpx, y, c.denoisingRange);
                        load_pair(cpx >> 1, y, vn[n], v1n[n]);
                        ok[n] = ok[n] && !gn.sky;
                        float w = smoothstep01(1.0f - absf(gn.z - g.z) * invDz);
                        wn[n] = ok[n] ? w : 0.0f;
                    }
                    if (!(wn[0] + wn[1] > 0.0f)) { // depth edge on both sides: plain mean of whatever exists
                        wn[0] = ok[0] ? 1.0f : 0.0f;
                        wn[1] = ok[1] ? 1.0f : 0.0f;
                    }
                    float wsum = wn[0] + wn[1];
                    f4 acc = wn[0] > 0.0f ? mul4(vn[0], wn[0]) : f4{0, 0, 0, 0};
                    acc = wn[1] > 0.0f ? fma4(vn[1], wn[1], acc) : acc;
                    f4 acc1 = wn[0] > 0.0f ? mul4(v1n[0], wn[0]) : f4{0, 0, 0, 0};
                    acc1 = wn[1] > 0.0f ? fma4(v1n[1], wn[1], acc1) : acc1;
                    float inv = rcp_(wsum);
                    v = wsum > 0.0f ? mul4(acc, inv) : f4{0, 0, 0, 0};
                    v1 = wsum > 0.0f ? mul4(acc1, inv) : f4{0, 0, 0, 0};
                }
                if (m.radius > 0 && v.w == 0.0f) { // no hit distance: reconstruct it from the neighbourhood
                    PixelGeo pg = pixel_geo(c, g, x, gy0, s.planeDistanceSensitivity);
                    float rough = isSpec ? g.roughness : 1.0f;
                    uint32_t minMat = isSpec ? s.minMaterialForSpecular : s.minMaterialForDiffuse;
                    float angle = spec_lobe_half_angle(rough) * s.lobeAngleFraction;
                    float normalW = rcp_(fmax2(angle, NORMAL_ANGLE_MIN));
                    normalW *= strand_normal_relax(c, g.mat, absf(g.z)); // CommonSettings::strandMaterialID: thin strands relax the normal test
                    float normalW2 = nw_param(normalW);
                    float roughA = rcp_(lerpf(0.01f, 1.0f, sat(rough * s.roughnessFraction)));
                    float roughB = -rough * roughA;
                    float sum = 0.0f, wsum = 0.0f;
                    for (int j = -m.radius; j <= m.radius; j++)
                        for (int i = -m.radius; i <= m.radius; i++) {
                            if (i == 0 && j == 0)
                                continue;
                            int px = x + i, py = y + j, gy = py + c.yOff;
                            if (px < 0 || px >= c.W || gy < 0 || gy >= c.H || py < 0 || py >= c.resH)
                                continue;
                            if (!has_data(phase, px, gy, c.frameIndex))
                                continue;
                            Guide gs = load_guide(G, px, py, c.denoisingRange);
                            if (gs.sky || material_mismatch(g.mat, gs.mat, minMat))
                                continue;
                            f4 hv, hv1;
                            load_pair(m.checker ? px >> 1 : px, py, hv, hv1);
                            float h = hv.w;
                            if (!(h > 0.0f))
                                continue;
                            float w = geo_weight(pg, (float)px, (float)gy, gs.z);
                            w *= normal_weight(normal_dist2(g.nw, gs.nw), normalW2);
                            if (isSpec)
                                w *= smoothstep01(1.0f - absf(fma_(gs.roughness, roughA, roughB)));
                            sum = fma_(h, w, sum);
                            wsum += w;
                        }
                    if (wsum > 0.0f)
                        v.w = sum * (rcp_(wsum));
                }
                if (d.occlusion || d.dirOcc)
                    v = {v.w, 0.0f, 0.0f, v.w};
                st_h4(out, x, y, v);
                if (sh1)
                    st_h4(*out1, x, y, v1);
            }
        }
}

struct SpatialIO {
    const Plane* in[2]; // per signal slot
    int inOff[2];       // byte offset of the signal inside the texel
    const Plane* out[2];
    int outOff[2];
    int reach; // taps farther than this many pixels (rows or columns) from the centre are rejected
    const Plane* in1[2]; // SH mode, PrePass only: the separate IN_*_SH1 planes (internal planes keep SH1 at +8 in the texel)
};

// SH mode (REBLUR_*_SH / RELAX_*_SH, Source/NRDSample.cpp:464-476): every signal carries a second texel (SH1, the
// direction-weighted first-order band, Shaders/TraceOpaque.cs.hlsl:738-752) that is filtered with EXACTLY the weights of
// SH0; luma rescales (history clamping, stabilization) scale SH1.xyz by the same factor.
static inline f4 load_sh1(const SpatialIO& io, int sig, int x, int y, bool pre) {
    return pre ? ld_h4(*io.in1[sig], x, y, 0) : ld_h4(*io.in[sig], x, y, io.inOff[sig] + 8);
}

// --------------------------------------------------------------------------------------------------
// Tap texels (REBLUR radiance flavours: no SH, no OCCLUSION). What a Blur / PostBlur tap needs - depth, normal, roughness, material
// and the signal - sits in ONE 16-byte texel per signal, so a tap is one gather instead of two (guide + radiance): the spatial passes
// are bound by what moves through the texture path, not by arithmetic (profiles/r03_ab_setup_planes.txt).
//   w0 = viewZ rounded to 22 bits | roughness as the 10-bit code of IN_NORMAL_ROUGHNESS   (read back AS A FLOAT it is the depth, the
//        roughness code perturbing it by < 2^-13 relative - every consumer reads it that way, centre and taps alike)
//   w1 = normal x | y << 10 | z << 20, 10 bits per component (n = code * 2/1023 - 1, not re-normalised) | materialID << 30
//   w2, w3 = the signal {Y, Co | Cg, hitT} as 4 x fp16
// The guide part IS the pixel's 8-byte guide texel (orc_core.h): HistoryFix copies it in, Blur copies it through, and both passes take
// their CENTRE pixel's guide from the texel as well (they do not touch the guide plane). Precision: the normal arrives in IN_NORMAL_ROUGHNESS as a 10 + 10 bit
// octahedron, the roughness as 10 bits - the texel is as fine as the input; only the depth loses its 10 low mantissa bits.
// --------------------------------------------------------------------------------------------------
struct TapTexel {
    uint32_t w0, w1, w2, w3;
};
static inline Guide unpack_tap_guide(uint32_t w0, uint32_t w1, float range) { return decode_guide_words(w0, w1, range); }
static inline TapTexel ld_tap(const Plane& P, int x, int y) {
    TapTexel t;
    std::memcpy(&t, texel(P, x, y), 16);
    return t;
}
static inline void st_tap(const Plane& P, int x, int y, uint32_t w0, uint32_t w1, f4 v) {
    uint16_t h[4] = {f32_to_f16(clampf(v.x, -FP16_MAX, FP16_MAX)), f32_to_f16(clampf(v.y, -FP16_MAX, FP16_MAX)),
                     f32_to_f16(clampf(v.z, -FP16_MAX, FP16_MAX)), f32_to_f16(clampf(v.w, -FP16_MAX, FP16_MAX))};
    uint8_t* t = texel(P, x, y);
    std::memcpy(t, &w0, 4);
    std::memcpy(t + 4, &w1, 4);
    std::memcpy(t + 8, h, 8);
}
static inline f4 tap_signal(const TapTexel& t) {
    return {f16_to_f32((uint16_t)(t.w2 & 0xffffu)), f16_to_f32((uint16_t)(t.w2 >> 16)), f16_to_f32((uint16_t)(t.w3 & 0xffffu)), f16_to_f32((uint16_t)(t.w3 >> 16))};
}
static inline bool tap_texels(const DenoiserState& d) { return d.kind == Kind::REBLUR && !d.sh; } // (OCCLUSION signals travel as {h, 0, 0, h} internally: same planes)
// REBLUR radiance flavours: PrePass + TemporalAccumulation as one dispatch unless NRDHIP_FLAG_SEPARATE_PASSES (csrc/nrdhip.cpp fused_prepass)
static inline bool fused_prepass(const Instance& I, const DenoiserState& d) {
    return d.kind == Kind::REBLUR && !d.sh && !d.occlusion && !(I.flags & NRDHIP_FLAG_SEPARATE_PASSES);
}

void spatial_filter(Ctx& k, Variant variant, const SpatialIO& io, int y0, int y1) {
    const Consts& c = k.c;
    const nrd::ReblurSettings& s = k.d.reblur;
    const Plane& G = k.guide();
    const Plane& D1 = k.perm(P_DATA1_A + k.cur);
    const Plane& HT = k.trans(T_HITTRACK);
    const float* hp = &s.hitDistanceParameters.A;
    const bool relaxIn = k.d.kind == Kind::RELAX && variant == PRE; // RELAX inputs: linear RGB + world-space hit distance
    const bool occIn = k.d.occlusion && variant == PRE && !prepare_mode(k.d).any; // PrepareInputs already expanded them to {h, 0, 0, h}
    const bool sh = k.d.sh;
    const bool tap = variant != PRE && tap_texels(k.d); // Blur / PostBlur on tap texels: io.in[sig] (and Blur's io.out[sig]) are tap planes
    const Plane& TILES = k.trans(T_TILES);
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c.W; x++) {
            // a tile without geometry (ClassifyTiles): PrePass and PostBlur write nothing there - what they would write is read by nobody
            // who has not tested the guide first (csrc/nrd_reblur.hip k_spatial). Blur keeps the tap texels of the tile current (below).
            if ((variant == PRE || variant == POST) && *texel(TILES, x >> 4, y >> 4))
                continue;
            TapTexel ctap[2] = {};
            if (tap)
                for (int sig = 0; sig < k.d.nsig; sig++)
                    ctap[sig] = ld_tap(*io.in[sig], x, y);
            Guide g = tap ? unpack_tap_guide(ctap[0].w0, ctap[0].w1, c.denoisingRange) : load_guide(G, x, y, c.denoisingRange);
            if (g.sky) {
                for (int sig = 0; sig < k.d.nsig; sig++) {
                    if (tap && variant == BLUR) { // the guide part travels on (PostBlur takes its sky test from it)
                        st_tap(*io.out[sig], x, y, ctap[sig].w0, ctap[sig].w1, {0, 0, 0, 0});
                        continue;
                    }
                    st_h4(*io.out[sig], x, y, {0, 0, 0, 0}, io.outOff[sig]);
                    if (sh)
                        st_h4(*io.out[sig], x, y, {0, 0, 0, 0}, io.outOff[sig] + 8);
                }
                if (variant == PRE && k.d.hasSpec)
                    st_h(HT, x, y, 0.0f);
                continue;
            }
            int gy0 = y + c.yOff;
            PixelGeo pg = pixel_geo(c, g, x, gy0, s.planeDistanceSensitivity);
            f3 V = to_viewer(c, pg.Xv);
            // Poisson rotation: per frame for PrePass / PostBlur (neighbouring pixels then gather neighbouring texels), per 2x2 quad for Blur
            bool perPixel = variant == BLUR;
            uint32_t h = hash_px(perPixel ? (uint32_t)x >> BLUR_ROTATION_SHIFT : 0u, perPixel ? (uint32_t)gy0 >> BLUR_ROTATION_SHIFT : 0u, c.frameIndex, (uint32_t)variant + 1u); // one rotation per 2x2 quad
            float rc = c.rot[h & 63u][0], rs = c.rot[h & 63u][1];
            float diffA = 0.0f, specA = 0.0f;
            if (variant != PRE)
                unpack_data1(ld_u16(D1, x, y), diffA, specA);

            for (int sig = 0; sig < k.d.nsig; sig++) {
                bool isSpec = (sig == k.sigSpec()) && k.d.hasSpec;
                float rough = isSpec ? g.roughness : 1.0f;
                uint32_t minMat = isSpec ? s.minMaterialForSpecular : s.minMaterialForDiffuse;
                f4 center = tap ? tap_signal(ctap[sig]) : load_signal(*io.in[sig], x, y, io.inOff[sig], occIn);
                if (relaxIn && !RELAX_LINEAR_RGB)
                    center = rgb_to_ycocg4(center);
                f4 sum1 = sh ? load_sh1(io, sig, x, y, variant == PRE) : f4{0, 0, 0, 0};
                const f4 center1 = sum1;
                // the pass-independent part of the set-up (csrc/nrd_reblur.hip KernelUnit)
                const KernelUnit ku = kernel_unit(c, s, pg, g.z, V, rough, isSpec);
                float hitNorm = fma_(pg.absZ, hp[1], hp[0]) * ku.hitFactor;
                float hitDist = center.w * hitNorm;
                float hitDistFactor = sat(hitDist * rcp_(pg.frustumSize));
                float A = isSpec ? specA : diffA;
                float nonLin = variant == PRE ? 1.0f : rcp_(1.0f + A);
                float smc = ku.smc;
                float radius;
                if (variant == PRE) {
                    radius = (isSpec ? s.specularPrepassBlurRadius : s.diffusePrepassBlurRadius) * hitDistFactor * smc;
                } else {
                    float r = fma_(s.maxBlurRadius * lerpf(MIN_CONVERGED_RADIUS_SCALE, 1.0f, nonLin), lerpf(hitDistFactor, 1.0f, nonLin), s.minBlurRadius);
                    r *= variant == POST ? POST_BLUR_RADIUS_SCALE : 1.0f;
                    r *= smc;
                    radius = s.maxBlurRadius != 0.0f ? r : 0.0f;
                }
                f4 sum = center;
                float wsum = 1.0f;
                float minHitW = center.w; // (PrePass: hit distance tracking, in units of the signal's w channel until the end)
                if (radius > 0.0f) {
                    // pixel offsets per unit of the (rotated) Poisson coordinates
                    float jtx = ku.j[0] * radius, jty = ku.j[1] * radius;
                    float jbx = ku.j[2] * radius, jby = ku.j[3] * radius;
                    float angle = ku.angle0 * lerpf(s.lobeAngleFraction, 1.0f, nonLin);
                    float normalW = wrcp_(fmax2(angle, NORMAL_ANGLE_MIN)); // (weight-class from here on: orc_math.h NRD_HW_TRANSCENDENTALS)
                    normalW *= strand_normal_relax(c, g.mat, absf(g.z)); // CommonSettings::strandMaterialID: thin strands relax the normal test
                    float normalW2 = nw_param(normalW);
                    float hitScale = relaxIn ? wrcp_(fmax2(center.w, 1e-3f)) : 1.0f; // RELAX hit distances are world units: compare relatively
                    float hitA = hitScale * wrcp_(lerpf(1e-6f, 1.0f, fmin2(nonLin, smc))) * EXP_WEIGHT_SCALE; // (the exponent's scale folded in: exp_weight_prescaled)
                    float hitB = -center.w * hitA;
                    float roughA = ku.roughA;
                    float roughB = -rough * roughA;
                    if (variant == BLUR) { // per-pixel rotation folded into the Jacobian (J . R): the taps then are the unrotated disk
                        const float a = fma_(rc, jtx, rs * jbx), b = fma_(rc, jbx, -(rs * jtx));
                        const float cc = fma_(rc, jty, rs * jby), d = fma_(rc, jby, -(rs * jty));
                        jtx = a;
                        jbx = b;
                        jty = cc;
                        jby = d;
                    }
                    for (int t = 0; t < 8; t++) {
                        float ox = variant == BLUR ? g_poisson8[t][0] : fma_(g_poisson8[t][0], rc, -(g_poisson8[t][1] * rs));
                        float oy = variant == BLUR ? g_poisson8[t][1] : fma_(g_poisson8[t][0], rs, g_poisson8[t][1] * rc);
                        float fpx = floorf(fma_(ox, jtx, fma_(oy, jbx, (float)x + 0.5f)));
                        float fpy = floorf(fma_(ox, jty, fma_(oy, jby, (float)gy0 + 0.5f)));
                        // tap window: inside the frame, inside the held rows, within the hard reach of the pass
                        const int loX = std::max(x - io.reach, 0), hiX = std::min(x + io.reach, c.W - 1);
                        const int loY = std::max(gy0 - io.reach, std::max(c.yOff, 0)), hiY = std::min(gy0 + io.reach, std::min(c.yOff + c.resH, c.H) - 1);
                        bool valid = fpx >= (float)loX && fpx <= (float)hiX && fpy >= (float)loY && fpy <= (float)hiY;
                        // PrePass reads caller-owned inputs (garbage allowed on sky / outside the rect): rejected taps are skipped.
                        // Blur / PostBlur read internal planes (always finite): a rejected tap enters with weight 0, its texel
                        // fetched at the position clamped into the window - no per-component select in the kernels.
                        if (variant == PRE && !valid)
                            continue;
                        int px = (int)clampf(fpx, (float)loX, (float)hiX), py = (int)clampf(fpy, (float)loY, (float)hiY) - c.yOff;
                        TapTexel tt = {};
                        if (tap)
                            tt = ld_tap(*io.in[sig], px, py);
                        Guide gs = tap ? unpack_tap_guide(tt.w0, tt.w1, c.denoisingRange) : load_guide(G, px, py, c.denoisingRange);
                        valid = valid && !gs.sky && !material_mismatch(g.mat, gs.mat, minMat);
                        if (variant == PRE && !valid)
                            continue;
                        f4 sv = tap ? tap_signal(tt) : load_signal(*io.in[sig], px, py, io.inOff[sig], occIn);
                        if (relaxIn && !RELAX_LINEAR_RGB)
                            sv = rgb_to_ycocg4(sv);
                        float w = 0.0f;
                        if (valid) {
                            w = g_poisson8[t][2];
                            w *= geo_weight(pg, fpx, fpy, gs.z);
                            w *= normal_weight(normal_dist2(g.nw, gs.nw), normalW2);
                            if (tap) { // the tap's roughness stays a 10-bit code: the scale of its decode is folded into the per-pixel constant
                                if (isSpec)
                                    w *= smoothstep01(1.0f - absf(fma_((float)(tt.w0 & 1023u), roughA * (1.0f / 1023.0f), roughB)));
                            } else if (isSpec)
                                w *= smoothstep01(1.0f - absf(fma_(gs.roughness, roughA, roughB)));
                            w *= lerpf(s.minHitDistanceWeight, 1.0f, exp_weight_prescaled(fma_(sv.w, hitA, hitB)));
                        }
                        sum = fma4(sv, w, sum);
                        if (sh)
                            sum1 = fma4(load_sh1(io, sig, px, py, variant == PRE), w, sum1);
                        wsum += w;
                        if (w > 0.0f)
                            minHitW = fmin2(minHitW, sv.w);
                    }
                }
                float invw = wrcp_(wsum);
                f4 res = mul4(sum, invw), res1 = mul4(sum1, invw);
                // usePrepassOnlyForSpecularMotionEstimation: the specular signal passes through, only the tracked hit distance is filtered
                if (variant == PRE && isSpec && k.d.kind == Kind::REBLUR && s.usePrepassOnlyForSpecularMotionEstimation) {
                    res = center;
                    res1 = center1;
                }
                if (tap && variant == BLUR)
                    st_tap(*io.out[sig], x, y, ctap[sig].w0, ctap[sig].w1, res);
                else
                    st_h4(*io.out[sig], x, y, res, io.outOff[sig]);
                if (sh)
                    st_h4(*io.out[sig], x, y, res1, io.outOff[sig] + 8);
                if (variant == PRE && isSpec)
                    st_h(HT, x, y, minHitW * hitNorm);
            }
        }
}

// --------------------------------------------------------------------------------------------------
// Reprojection helpers shared by TemporalAccumulation and TemporalStabilization
// --------------------------------------------------------------------------------------------------
struct Reproj {
    float su, sv; // surface motion based previous uv
    f3 Xw;        // camera-relative world position
    f3 XwPrev;    // world position in the previous frame (relative to the CURRENT camera)
    f3 XvPrev;    // previous view position
    float zPrev;
};

static inline Reproj reproject(const Consts& c, f3 Xv, float u, float v, f4 mvRaw) {
    Reproj r;
    r.Xw = rot3(c.v2w, Xv);
    f3 mv = {mvRaw.x * c.mvScale[0], mvRaw.y * c.mvScale[1], mvRaw.z * c.mvScale[2]};
    f3 cd = {c.camDelta[0], c.camDelta[1], c.camDelta[2]};
    if (c.mvWorld) {
        r.XwPrev = add3(r.Xw, mv);
        r.XvPrev = rot3(c.w2vPrev, sub3(r.XwPrev, cd));
        r.zPrev = r.XvPrev.z;
        if (!project(c.pjPrev, r.XvPrev, r.su, r.sv)) {
            r.su = -10.0f;
            r.sv = -10.0f;
        }
    } else {
        r.su = u + mv.x;
        r.sv = v + mv.y;
        if (c.mvScale[2] != 0.0f) {
            r.zPrev = Xv.z + mv.z;
            r.XvPrev = reconstruct(c.frPrev, r.su, r.sv, r.zPrev);
            r.XwPrev = add3(rot3(c.v2wPrev, r.XvPrev), cd); // relative to the current camera
        } else {
            r.XwPrev = r.Xw; // static point
            r.XvPrev = rot3(c.w2vPrev, sub3(r.XwPrev, cd));
            r.zPrev = r.XvPrev.z;
        }
    }
    return r;
}

struct Footprint {
    int ix, iy; // global coordinates of tap (0,0)
    float w[4]; // bilinear weights * validity, order (0,0) (1,0) (0,1) (1,1)
    float wsum;
    uint32_t bits;
};

// bilinear footprint in the previous frame with per-tap occlusion test against the plane (NvPrev, XvPrev)
static inline Footprint footprint(const Ctx& k, float pu, float pv, f3 NvPrev, f3 XvPrev, f3 N, uint32_t mat, uint32_t minMat, float threshold) {
    const Consts& c = k.c;
    const Plane& GP = k.guidePrev();
    Footprint f;
    float px = fma_(pu, (float)c.Wprev, -0.5f), py = fma_(pv, (float)c.Hprev, -0.5f);
    float fx0 = floorf(px), fy0 = floorf(py);
    float fx = px - fx0, fy = py - fy0;
    bool sane = fx0 >= -2.0f && fx0 <= (float)c.Wprev + 1.0f && fy0 >= -2.0f && fy0 <= (float)c.Hprev + 1.0f;
    f.ix = sane ? (int)fx0 : -4;
    f.iy = sane ? (int)fy0 : -4;
    float bw[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
    f.wsum = 0.0f;
    f.bits = 0;
    float planeRef = dot3(NvPrev, XvPrev);
    float g0 = c.ortho ? fma_(NvPrev.x, c.pvPrev[0], NvPrev.y * c.pvPrev[1]) : fma_(NvPrev.x, c.pvPrev[0], fma_(NvPrev.y, c.pvPrev[1], NvPrev.z));
    float gx = NvPrev.x * c.pvPrev[2], gyc = NvPrev.y * c.pvPrev[3];
    for (int i = 0; i < 4; i++) {
        int tx = f.ix + (i & 1), gy = f.iy + (i >> 1), ty = gy - c.yOff;
        bool ok = sane && tx >= 0 && tx < c.Wprev && gy >= 0 && gy < c.Hprev && ty >= c.prevY0 && ty < c.prevY1;
        if (ok) {
            Guide gp = load_guide(GP, tx, ty, c.denoisingRange);
            float lin = fma_(gx, (float)tx, fma_(gyc, (float)gy, g0));
            float plane = c.ortho ? fma_(gp.z, NvPrev.z, lin) : gp.z * lin; // N . X of the previous-frame texel
            ok = !gp.sky && absf(plane - planeRef) <= threshold && dot3(N, gp.n) > PREV_NORMAL_COS && !material_mismatch(mat, gp.mat, minMat);
        }
        f.w[i] = ok ? bw[i] : 0.0f;
        f.wsum += f.w[i];
        f.bits |= ok ? (1u << i) : 0u;
    }
    return f;
}

static inline f4 fetch4(const Ctx& k, const Plane& P, int off, const Footprint& f) {
    f4 s = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        if (f.w[i] > 0.0f)
            s = fma4(ld_h4(P, f.ix + (i & 1), f.iy + (i >> 1) - k.c.yOff, off), f.w[i], s);
    return mul4(s, rcp_(f.wsum));
}
static inline float fetch1(const Ctx& k, const Plane& P, int off, const Footprint& f) {
    float s = 0.0f;
    for (int i = 0; i < 4; i++)
        if (f.w[i] > 0.0f)
            s = fma_(ld_h(P, f.ix + (i & 1), f.iy + (i >> 1) - k.c.yOff, off), f.w[i], s);
    return s * (rcp_(f.wsum));
}
static inline void fetchA(const Ctx& k, const Plane& P, const Footprint& f, float& dA, float& sA) {
    dA = sA = 0.0f;
    for (int i = 0; i < 4; i++)
        if (f.w[i] > 0.0f) {
            float a, b;
            unpack_data1(ld_u16(P, f.ix + (i & 1), f.iy + (i >> 1) - k.c.yOff), a, b);
            dA = fma_(a, f.w[i], dA);
            sA = fma_(b, f.w[i], sA);
        }
    float inv = rcp_(f.wsum);
    dA *= inv;
    sA *= inv;
}

// virtual-motion previous uv of the specular reflection (shared by TA and TS). Surfaces of
// CommonSettings::cameraAttachedReflectionMaterialID (Source/NRDSample.cpp:3869-3876) reflect objects that travel with the
// camera: their virtual point keeps its VIEW-space position, so it is projected as it stands in the current view.
static inline bool virtual_uv(const Consts& c, const Reproj& r, float hitDist, float roughness, uint32_t mat, float& vu, float& vv) {
    f3 toCam = c.ortho ? rot3(c.v2w, f3{0.0f, 0.0f, r.zPrev >= 0.0f ? 1.0f : -1.0f}) : normalize3(r.Xw); // direction camera -> surface
    float f = spec_dominant_factor(roughness);
    f3 Xvirt = add3(r.Xw, mul3(toCam, hitDist * f));
    f3 XvirtPrev = add3(Xvirt, sub3(r.XwPrev, r.Xw));
    f3 rel = sub3(XvirtPrev, {c.camDelta[0], c.camDelta[1], c.camDelta[2]});
    f3 Xp = rot3(c.w2vPrev, rel);
    if (c.camAttachMat != 0xffffffffu && mat == c.camAttachMat)
        Xp = rot3(c.w2v, Xvirt);
    return project(c.pjPrev, Xp, vu, vv);
}

// bilinear sample of the (any-resolution) confidence plane, channel x
static inline float sample_confidence(const Plane& P, float u, float v) {
    if (!P.p)
        return 1.0f;
    float px = fma_(u, (float)P.w, -0.5f), py = fma_(v, (float)P.h, -0.5f);
    float fx0 = floorf(px), fy0 = floorf(py);
    float fx = px - fx0, fy = py - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    auto at = [&](int x, int y) {
        x = x < 0 ? 0 : (x >= P.w ? P.w - 1 : x);
        y = y < 0 ? 0 : (y >= P.h ? P.h - 1 : y);
        return ld_h(P, x, y, 0);
    };
    float a = lerpf(at(x0, y0), at(x0 + 1, y0), fx);
    float b = lerpf(at(x0, y0 + 1), at(x0 + 1, y0 + 1), fx);
    return sat(lerpf(a, b, fy));
}

// surface-motion specular accumulation limit under parallax
static inline float spec_accum_limit(float roughness, float NoV, float parallaxPx) {
    float acos01sq = sat(1.0f - NoV * 0.99999f);
    float a = sqrt_(acos01sq);
    float b = fma_(roughness, roughness, 1.1f);
    float parallaxSensitivity = (b + a) * rcp_(b - a);
    float powerScale = fma_(parallaxSensitivity * parallaxPx, 2.0f, 1.0f);
    float f = 1.0f - exp2_poly(-200.0f * roughness * roughness);
    f *= pow01(roughness, 0.5f * powerScale);
    return MAX_ACCUM * f;
}

// --------------------------------------------------------------------------------------------------
// K3 TemporalAccumulation
// --------------------------------------------------------------------------------------------------
// `prepassResult`: the plane the PrePass wrote its result to - Tmp1, or the private scratch rows of the fused dispatch (below)
static void temporal_accumulation_rows(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1, const Plane* prepassResult);
void temporal_accumulation(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) { temporal_accumulation_rows(I, d, c, y0, y1, nullptr); }
static void temporal_accumulation_rows(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1, const Plane* prepassResult) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const int sb = d.sh ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    (void)sb;
    const nrd::ReblurSettings& s = d.reblur;
    const Plane& G = k.guide();
    const Plane& MV = k.slot(nrd::ResourceType::IN_MV);
    const Plane& IN = prepassResult ? *prepassResult : k.trans(T_TMP1);
    const Plane& OUT = k.trans(T_TMP2);
    const Plane& HIST = k.perm(P_HIST);
    const Plane& FASTP = k.perm(P_FAST_A + (k.cur ^ 1));
    const Plane& FASTC = k.perm(P_FAST_A + k.cur);
    const Plane& D1P = k.perm(P_DATA1_A + (k.cur ^ 1));
    const Plane& D1T = k.trans(T_DATA1);
    const Plane& D2 = k.trans(T_DATA2);
    const Plane& HT = k.trans(T_HITTRACK);
    const Plane& confD = k.slot(nrd::ResourceType::IN_DIFF_CONFIDENCE);
    const Plane& confS = k.slot(nrd::ResourceType::IN_SPEC_CONFIDENCE);
    bool historyOk = d.historyValid && !c.reset;
    const bool relax = d.kind == Kind::RELAX;
    float maxA = (float)std::min<uint32_t>(s.maxAccumulatedFrameNum, 63);
    float maxFastA = (float)std::min<uint32_t>(s.maxFastAccumulatedFrameNum, 63);
    float maxAs = relax ? (float)std::min<uint32_t>(d.relax.specularMaxAccumulatedFrameNum, 63) : maxA;
    float maxFastAs = relax ? (float)std::min<uint32_t>(d.relax.specularMaxFastAccumulatedFrameNum, 63) : maxFastA;
    // RELAX: second luma moment history lives in the slots REBLUR uses for the stabilized luma
    const Plane& MOMP = k.perm(P_STAB_A + (k.cur ^ 1));
    const Plane& MOMC = k.perm(P_STAB_A + k.cur);
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c.W; x++) {
            if (*texel(k.trans(T_TILES), x >> 4, y >> 4)) // a tile without geometry: nothing to write (csrc/nrd_reblur.hip k_temporal_accumulation)
                continue;
            Guide g = load_guide(G, x, y, c.denoisingRange);
            if (g.sky) {
                for (int sig = 0; sig < d.nsig; sig++) {
                    st_h4(OUT, x, y, {0, 0, 0, 0}, sig * sb);
                    if (d.sh)
                        st_h4(OUT, x, y, {0, 0, 0, 0}, sig * sb + 8);
                    st_h(FASTC, x, y, 0.0f, sig * 2);
                    if (relax)
                        st_h(MOMC, x, y, 0.0f, sig * 2);
                }
                st_u16(D1T, x, y, 0);
                st_u32(D2, x, y, 0);
                continue;
            }
            int gy0 = y + c.yOff;
            float u = ((float)x + 0.5f) * c.invW, v = ((float)gy0 + 0.5f) * c.invH;
            f3 Xv = reconstruct_px(c.pv, (float)x, (float)gy0, g.z);
            f3 Nv = rot3(c.w2v, g.n);
            f3 V = to_viewer(c, Xv);
            float NoV = absf(dot3(Nv, V));
            Reproj r = reproject(c, Xv, u, v, ld_h4(MV, x, y));
            f3 NvPrev = rot3(c.w2vPrev, g.n);
            float thrBase = c.disocclusionThreshold;
            if (c.mixAvail) // per-pixel blend toward disocclusionThresholdAlternate (IN_DISOCCLUSION_THRESHOLD_MIX, R8_UNORM)
                thrBase = lerpf(thrBase, c.disoccAlt, (float)*texel(k.slot(nrd::ResourceType::IN_DISOCCLUSION_THRESHOLD_MIX), x, y) * (1.0f / 255.0f));
            float threshold = thrBase * c.minRectDimMulUnproject * (c.ortho ? 1.0f : absf(r.zPrev));
            uint32_t minMatAny = std::min<uint32_t>(s.minMaterialForDiffuse, s.minMaterialForSpecular);
            Footprint smb = footprint(k, r.su, r.sv, NvPrev, r.XvPrev, g.n, g.mat, minMatAny, threshold);
            bool smbOk = historyOk && smb.wsum > 0.0f;
            float prevDiffA = 0.0f, prevSpecA = 0.0f;
            if (smbOk)
                fetchA(k, D1P, smb, prevDiffA, prevSpecA);
            // "+1": the stored value is the accumulation speed the previous frame USED
            prevDiffA = smbOk ? fmin2(prevDiffA + 1.0f, maxA) : 0.0f;
            prevSpecA = smbOk ? fmin2(prevSpecA + 1.0f, maxAs) : 0.0f;
            float quality = smbOk ? smb.wsum : 0.0f;
            float outDiffA = 0.0f, outSpecA = 0.0f;
            uint32_t data2 = smbOk ? smb.bits : 0u;

            if (d.hasDiff) {
                int sig = k.sigDiff();
                f4 in = ld_h4(IN, x, y, sig * sb);
                float A = prevDiffA;
                if (c.confAvail)
                    A *= sample_confidence(confD, u, v);
                A *= lerpf(quality, 1.0f, rcp_(1.0f + A));
                float nonLin = rcp_(1.0f + A);
                f4 hist = smbOk ? fetch4(k, HIST, sig * sb, smb) : in;
                float fastHist = smbOk ? fetch1(k, FASTP, sig * 2, smb) : signal_luma(in, relax);
                st_h4(OUT, x, y, lerp4(hist, in, nonLin), sig * sb);
                if (d.sh) { // SH1 follows SH0: same footprint, same blend factor
                    f4 in1 = ld_h4(IN, x, y, sig * sb + 8);
                    f4 hist1 = smbOk ? fetch4(k, HIST, sig * sb + 8, smb) : in1;
                    st_h4(OUT, x, y, lerp4(hist1, in1, nonLin), sig * sb + 8);
                }
                st_h(FASTC, x, y, lerpf(fastHist, signal_luma(in, relax), rcp_(1.0f + fmin2(A, maxFastA))), sig * 2);
                if (relax) {
                    float m2 = signal_luma(in, relax) * signal_luma(in, relax);
                    float m2prev = smbOk ? fetch1(k, MOMP, sig * 2, smb) : m2;
                    st_h(MOMC, x, y, lerpf(m2prev, m2, nonLin), sig * 2);
                }
                outDiffA = A;
            }
            if (d.hasSpec) {
                int sig = k.sigSpec();
                f4 in = ld_h4(IN, x, y, sig * sb);
                float hitDist = ld_h(HT, x, y);
                // parallax (pixels) of the point seen from the previous camera position
                f3 Xpar = sub3(r.XwPrev, {c.camDelta[0], c.camDelta[1], c.camDelta[2]});
                f3 XparV = rot3(c.w2v, Xpar);
                float pu, pv, parallax = 0.0f;
                if (project(c.pj, XparV, pu, pv)) {
                    float dx = (pu - r.su) * (float)c.W, dy = (pv - r.sv) * (float)c.H;
                    parallax = sqrt_(fma_(dx, dx, dy * dy));
                }
                float Asmb = fmin2(prevSpecA, spec_accum_limit(g.roughness, NoV, parallax));
                // virtual motion
                float vu, vv;
                float amount = 0.0f, Avmb = 0.0f;
                f4 vmbHist = in;
                float vmbFast = signal_luma(in, relax);
                Footprint vmb;
                vmb.bits = 0;
                vmb.wsum = 0.0f;
                if (historyOk && virtual_uv(c, r, hitDist, g.roughness, g.mat, vu, vv)) {
                    vmb = footprint(k, vu, vv, NvPrev, r.XvPrev, g.n, g.mat, s.minMaterialForSpecular, threshold);
                    if (vmb.wsum > 0.0f) {
                        // roughness similarity of the virtual footprint
                        float prevRough = 0.0f;
                        for (int i = 0; i < 4; i++)
                            if (vmb.w[i] > 0.0f)
                                prevRough = fma_(guide_roughness(k.guidePrev(), vmb.ix + (i & 1), vmb.iy + (i >> 1) - c.yOff), vmb.w[i], prevRough);
                        prevRough *= rcp_(vmb.wsum);
                        float roughA = rcp_(lerpf(0.01f, 1.0f, sat(g.roughness * s.roughnessFraction)));
                        float rconf = smoothstep01(1.0f - absf((prevRough - g.roughness) * roughA));
                        amount = spec_dominant_factor(g.roughness) * vmb.wsum * rconf;
                        float dA, sA;
                        fetchA(k, D1P, vmb, dA, sA);
                        Avmb = fmin2(sA + 1.0f, maxAs);
                        vmbHist = fetch4(k, HIST, sig * sb, vmb);
                        vmbFast = fetch1(k, FASTP, sig * 2, vmb);
                    }
                }
                f4 smbHist = smbOk ? fetch4(k, HIST, sig * sb, smb) : in;
                float smbFast = smbOk ? fetch1(k, FASTP, sig * 2, smb) : signal_luma(in, relax);
                if (!smbOk)
                    Asmb = 0.0f;
                float A = lerpf(Asmb, Avmb, amount);
                if (c.confAvail)
                    A *= sample_confidence(confS, u, v);
                float q = lerpf(quality, 1.0f, amount);
                A *= lerpf(q, 1.0f, rcp_(1.0f + A));
                // responsive accumulation for very smooth surfaces
                if (s.responsiveAccumulationSettings.roughnessThreshold > 0.0f) {
                    float t = smoothstep01(g.roughness / s.responsiveAccumulationSettings.roughnessThreshold);
                    A = fmin2(A, lerpf((float)s.responsiveAccumulationSettings.minAccumulatedFrameNum, maxAs, t));
                }
                float nonLin = rcp_(1.0f + A);
                f4 hist = lerp4(smbHist, vmbHist, amount);
                float fastHist = lerpf(smbFast, vmbFast, amount);
                st_h4(OUT, x, y, lerp4(hist, in, nonLin), sig * sb);
                if (d.sh) {
                    f4 in1 = ld_h4(IN, x, y, sig * sb + 8);
                    f4 smb1 = smbOk ? fetch4(k, HIST, sig * sb + 8, smb) : in1;
                    f4 vmb1 = vmb.wsum > 0.0f ? fetch4(k, HIST, sig * sb + 8, vmb) : in1;
                    st_h4(OUT, x, y, lerp4(lerp4(smb1, vmb1, amount), in1, nonLin), sig * sb + 8);
                }
                st_h(FASTC, x, y, lerpf(fastHist, signal_luma(in, relax), rcp_(1.0f + fmin2(A, maxFastAs))), sig * 2);
                if (relax) {
                    float m2 = signal_luma(in, relax) * signal_luma(in, relax);
                    float m2smb = smbOk ? fetch1(k, MOMP, sig * 2, smb) : m2;
                    float m2vmb = vmb.wsum > 0.0f ? fetch1(k, MOMP, sig * 2, vmb) : m2;
                    st_h(MOMC, x, y, lerpf(lerpf(m2smb, m2vmb, amount), m2, nonLin), sig * 2);
                }
                outSpecA = A;
                // bits 16..23: reprojection confidence of the specular history (read by the RELAX A-trous edge-stopping relaxation)
                data2 |= (vmb.bits << 4) | ((uint32_t)floorf(fma_(sat(amount), 255.0f, 0.5f)) << 8) | (relax ? (uint32_t)floorf(fma_(sat(q), 255.0f, 0.5f)) << 16 : 0u);
            }
            st_u16(D1T, x, y, pack_data1(outDiffA, outSpecA));
            st_u32(D2, x, y, data2);
        }
}

// --------------------------------------------------------------------------------------------------
// K4 HistoryFix: sparse 5x5 reconstruction for short histories + fast-history clamping
// --------------------------------------------------------------------------------------------------
void history_fix(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const int sb = d.sh ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    (void)sb;
    const nrd::ReblurSettings& s = d.reblur;
    const Plane& G = k.guide();
    const Plane& IN = k.trans(T_TMP2);
    const bool relax = d.kind == Kind::RELAX;
    const Plane& OUT = relax ? k.perm(P_HIST) : k.trans(T_TMP1); // RELAX: the fixed + clamped signal IS the next frame's history
    const bool tap = tap_texels(d); // REBLUR radiance flavours: the result goes out as tap texels (guide + signal) for the Blur
    const Plane& FAST = k.perm(P_FAST_A + k.cur);
    const Plane& D1T = k.trans(T_DATA1);
    const Plane& D1C = k.perm(P_DATA1_A + k.cur);
    const Plane& MOM = k.perm(P_STAB_A + k.cur); // RELAX: accumulated second luma moment (antilag)
    const nrd::RelaxAntilagSettings& al = d.relax.antilagSettings;
    float maxFastAd = (float)std::min<uint32_t>(s.maxFastAccumulatedFrameNum, 63);
    float maxFastAs = relax ? (float)std::min<uint32_t>(d.relax.specularMaxFastAccumulatedFrameNum, 63) : maxFastAd;
    bool clampEnabled = s.maxFastAccumulatedFrameNum < s.maxAccumulatedFrameNum;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c.W; x++) {
            Guide g = load_guide(G, x, y, c.denoisingRange);
            const uint32_t tw0 = ld_u32(G, x, y, 0), tw1 = ld_u32(G, x, y, 4); // the tap texels carry the guide texel as it is
            if (g.sky) {
                for (int sig = 0; sig < d.nsig; sig++) {
                    if (tap) {
                        st_tap(k.trans(T_TAP_D_A + ((sig == k.sigSpec() && d.hasSpec) ? 1 : 0)), x, y, tw0, tw1, {0, 0, 0, 0});
                        continue;
                    }
                    st_h4(OUT, x, y, {0, 0, 0, 0}, sig * sb);
                    if (d.sh)
                        st_h4(OUT, x, y, {0, 0, 0, 0}, sig * sb + 8);
                }
                st_u16(D1C, x, y, 0);
                continue;
            }
            int gy0 = y + c.yOff;
            float A[2];
            unpack_data1(ld_u16(D1T, x, y), A[0], A[1]);
            float outA[2] = {A[0], A[1]};
            bool geoReady = false;
            PixelGeo pg;
            for (int sig = 0; sig < d.nsig; sig++) {
                bool isSpec = (sig == k.sigSpec()) && d.hasSpec;
                int ai = isSpec ? 1 : 0;
                float rough = isSpec ? g.roughness : 1.0f;
                uint32_t minMat = isSpec ? s.minMaterialForSpecular : s.minMaterialForDiffuse;
                f4 val = ld_h4(IN, x, y, sig * sb);
                f4 val1 = d.sh ? ld_h4(IN, x, y, sig * sb + 8) : f4{0, 0, 0, 0};
                float Acur = A[ai];
                // ---- history reconstruction
                if (Acur < (float)s.historyFixFrameNum && s.historyFixFrameNum > 0) {
                    float normA = sat(Acur / (float)s.historyFixFrameNum);
                    int stride = (int)floorf(fma_((float)s.historyFixBasePixelStride, 1.0f - normA, 0.5f));
                    if (stride > 0) {
                        if (!geoReady) {
                            pg = pixel_geo(c, g, x, gy0, s.planeDistanceSensitivity);
                            geoReady = true;
                        }
                        float angle = spec_lobe_half_angle(rough) * lerpf(s.lobeAngleFraction, 1.0f, rcp_(1.0f + Acur));
                        float normalW = rcp_(fmax2(angle, NORMAL_ANGLE_MIN));
                        normalW *= strand_normal_relax(c, g.mat, absf(g.z)); // CommonSettings::strandMaterialID: thin strands relax the normal test
                        float normalW2 = nw_param(normalW);
                        float roughA = rcp_(lerpf(0.01f, 1.0f, sat(rough * s.roughnessFraction)));
                        float roughB = -rough * roughA;
                        f4 sum = mul4(val, 1.0f + Acur);
                        f4 sum1 = mul4(val1, 1.0f + Acur);
                        float wsum = 1.0f + Acur;
                        for (int j = -2; j <= 2; j++)
                            for (int i = -2; i <= 2; i++) {
                                if ((i == 0 && j == 0) || (i * i == 4 && j * j == 4))
                                    continue;
                                int px = x + i * stride, py = y + j * stride, gy = py + c.yOff;
                                if (px < 0 || px >= c.W || gy < 0 || gy >= c.H || py < 0 || py >= c.resH)
                                    continue;
                                Guide gs = load_guide(G, px, py, c.denoisingRange);
                                if (gs.sky || material_mismatch(g.mat, gs.mat, minMat))
                                    continue;
                                float w = rcp_(1.0f + (float)(i * i + j * j));
                                w *= geo_weight(pg, (float)px, (float)gy, gs.z);
                                // RELAX: pow(N.Ns, historyFixEdgeStoppingNormalPower) (sample UI Source/NRDSample.cpp:1626) instead of the lobe weight
                                w *= relax ? pow01(normal_cos(g.nw, gs.nw), d.relax.historyFixEdgeStoppingNormalPower) : normal_weight(normal_dist2(g.nw, gs.nw), normalW2);
                                if (isSpec)
                                    w *= smoothstep01(1.0f - absf(fma_(gs.roughness, roughA, roughB)));
                                float tA[2];
                                unpack_data1(ld_u16(D1T, px, py), tA[0], tA[1]);
                                w *= 1.0f + tA[ai];
                                sum = fma4(ld_h4(IN, px, py, sig * sb), w, sum);
                                if (d.sh)
                                    sum1 = fma4(ld_h4(IN, px, py, sig * sb + 8), w, sum1);
                                wsum += w;
                            }
                        val = mul4(sum, rcp_(wsum));
                        val1 = mul4(sum1, rcp_(wsum));
                    }
                }
                // ---- fast history clamping (5x5 moments of the fast luma history)
                if (clampEnabled) {
                    float fc = ld_h(FAST, x, y, sig * 2);
                    float m1 = 0.0f, m2 = 0.0f;
                    for (int j = -2; j <= 2; j++)
                        for (int i = -2; i <= 2; i++) {
                            int px = x + i, py = y + j, gy = py + c.yOff;
                            float f = fc;
                            if (px >= 0 && px < c.W && gy >= 0 && gy < c.H && py >= 0 && py < c.resH) {
                                float zt = ld_f32(G, px, py, 0);
                                if (absf(zt) <= c.denoisingRange)
                                    f = ld_h(FAST, px, py, sig * 2);
                            }
                            m1 += f;
                            m2 = fma_(f, f, m2);
                        }
                    m1 *= 1.0f / 25.0f;
                    m2 *= 1.0f / 25.0f;
                    float sigma = sqrt_(fmax2(fma_(-m1, m1, m2), 0.0f)) * s.fastHistoryClampingSigmaScale;
                    float Y = signal_luma(val, relax);
                    float Yc = clampf(Y, m1 - sigma, m1 + sigma);
                    float scale = (Yc + 1e-6f) * rcps_(Y + 1e-6f);
                    clamp_luma(val, Yc, scale, relax);
                    val1.x *= scale;
                    val1.y *= scale;
                    val1.z *= scale;
                    float f = sat(absf(Yc - Y) * rcp_(fmax2(fmax2(Y, Yc), 1e-6f)));
                    // RELAX antilag (RelaxAntilagSettings, sample UI Source/NRDSample.cpp:1600-1606): a clamped pixel accelerates its
                    // history by accelerationAmount; a history further from the fast 5x5 mean than spatialSigmaScale x spatial sigma +
                    // temporalSigmaScale x temporal sigma is reset by up to resetAmount (ramp over one more threshold)
                    if (relax)
                        f *= sat(al.accelerationAmount);
                    outA[ai] = lerpf(Acur, fmin2(Acur, isSpec ? maxFastAs : maxFastAd), f);
                    if (relax) {
                        float sigS = sqrt_(fmax2(fma_(-m1, m1, m2), 0.0f));
                        float sigT = sqrt_(fmax2(fma_(-Y, Y, ld_h(MOM, x, y, sig * 2)), 0.0f));
                        float thr = fma_(al.spatialSigmaScale, sigS, al.temporalSigmaScale * sigT);
                        float over = sat(fma_(absf(Y - m1), rcp_(fmax2(thr, 1e-6f)), -1.0f));
                        outA[ai] *= fma_(-sat(al.resetAmount), over, 1.0f);
                    }
                }
                // ---- anti-firefly (enableAntiFirefly, sample UI Source/NRDSample.cpp:1515-1582): luma clamped to the moments of the
                // 5x5 neighbourhood of the incoming signal WITHOUT its centre, chroma (and SH1) re-scaled with it
                if (s.enableAntiFirefly) {
                    float cc = ld_luma(IN, x, y, sig * sb, relax);
                    float m1 = 0.0f, m2 = 0.0f;
                    for (int j = -2; j <= 2; j++)
                        for (int i = -2; i <= 2; i++) {
                            if (i == 0 && j == 0)
                                continue;
                            int px = x + i, py = y + j, gy = py + c.yOff;
                            float f = cc;
                            if (px >= 0 && px < c.W && gy >= 0 && gy < c.H && py >= 0 && py < c.resH) {
                                float zt = ld_f32(G, px, py, 0);
                                if (absf(zt) <= c.denoisingRange)
                                    f = ld_luma(IN, px, py, sig * sb, relax);
                            }
                            m1 += f;
                            m2 = fma_(f, f, m2);
                        }
                    m1 *= 1.0f / 24.0f;
                    m2 *= 1.0f / 24.0f;
                    float sigma = sqrt_(fmax2(fma_(-m1, m1, m2), 0.0f)) * s.fireflySuppressorMinRelativeScale;
                    float Y = signal_luma(val, relax);
                    float Yc = clampf(Y, m1 - sigma, m1 + sigma);
                    float scale = (Yc + 1e-6f) * rcps_(Y + 1e-6f);
                    clamp_luma(val, Yc, scale, relax);
                    val1.x *= scale;
                    val1.y *= scale;
                    val1.z *= scale;
                }
                if (tap)
                    st_tap(k.trans(T_TAP_D_A + (isSpec ? 1 : 0)), x, y, tw0, tw1, val);
                else
                    st_h4(OUT, x, y, val, sig * sb);
                if (d.sh)
                    st_h4(OUT, x, y, val1, sig * sb + 8);
            }
            st_u16(D1C, x, y, pack_data1(outA[0], outA[1]));
        }
}

// --------------------------------------------------------------------------------------------------
// K7 TemporalStabilization (+ split screen)
// --------------------------------------------------------------------------------------------------
void temporal_stabilization(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const int sb = d.sh ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    (void)sb;
    const nrd::ReblurSettings& s = d.reblur;
    const Plane& G = k.guide();
    const Plane& MV = k.slot(nrd::ResourceType::IN_MV);
    const Plane& HIST = k.perm(P_HIST);
    const Plane& STABP = k.perm(P_STAB_A + (k.cur ^ 1));
    const Plane& STABC = k.perm(P_STAB_A + k.cur);
    const Plane& D1 = k.perm(P_DATA1_A + k.cur);
    const Plane& D2 = k.trans(T_DATA2);
    const Plane& HT = k.trans(T_HITTRACK);
    const Plane* outP[2] = {nullptr, nullptr};
    // split screen shows the noisy input: the slot itself, or its dense PrepareInputs copy when that pass ran
    const PrepareMode pm = prepare_mode(d);
    const bool occIn = d.occlusion && !pm.any;
    const Plane* inP[2] = {nullptr, nullptr};
    const Plane* out1P[2] = {nullptr, nullptr};
    const Plane* in1P[2] = {nullptr, nullptr};
    if (d.hasDiff) {
        outP[k.sigDiff()] = &k.slot(out_slot(d, false));
        inP[k.sigDiff()] = pm.any ? &k.trans(T_PREP_D) : &k.slot(in_slot(d, false));
        out1P[k.sigDiff()] = &k.slot(out1_slot(false));
        in1P[k.sigDiff()] = pm.sh1 ? &k.trans(T_PREP_D1) : &k.slot(in1_slot(false));
    }
    const bool dirOcc = d.dirOcc; // single {SH1.xyz, SH0.x} texel out, raw texel copy left of the split screen
    if (d.hasSpec) {
        outP[k.sigSpec()] = &k.slot(out_slot(d, true));
        inP[k.sigSpec()] = pm.any ? &k.trans(T_PREP_S) : &k.slot(in_slot(d, true));
        out1P[k.sigSpec()] = &k.slot(out1_slot(true));
        in1P[k.sigSpec()] = pm.sh1 ? &k.trans(T_PREP_S1) : &k.slot(in1_slot(true));
    }
    bool historyOk = d.historyValid && !c.reset;
    float maxStab = (float)std::min<uint32_t>(s.maxStabilizedFrameNum, 63);
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c.W; x++) {
            int gy0 = y + c.yOff;
            float u = ((float)x + 0.5f) * c.invW, v = ((float)gy0 + 0.5f) * c.invH;
            bool split = u < c.splitScreen;
            Guide g = load_guide(G, x, y, c.denoisingRange);
            if (g.sky) {
                for (int sig = 0; sig < d.nsig; sig++) {
                    if (dirOcc)
                        st_dir(*outP[sig], x, y, split ? dir_pass(*inP[sig], *in1P[sig], x, y) : f4{0, 0, 0, 0});
                    else
                        store_signal(*outP[sig], x, y, split ? load_signal(*inP[sig], x, y, 0, occIn) : f4{0, 0, 0, 0}, d.occlusion);
                    if (d.sh && !dirOcc)
                        st_h4(*out1P[sig], x, y, split ? ld_h4(*in1P[sig], x, y) : f4{0, 0, 0, 0});
                    st_h(STABC, x, y, 0.0f, sig * 2);
                }
                continue;
            }
            f3 Xv = reconstruct_px(c.pv, (float)x, (float)gy0, g.z);
            Reproj r = reproject(c, Xv, u, v, ld_h4(MV, x, y));
            uint32_t data2 = ld_u32(D2, x, y);
            float A[2];
            unpack_data1(ld_u16(D1, x, y), A[0], A[1]);
            for (int sig = 0; sig < d.nsig; sig++) {
                bool isSpec = (sig == k.sigSpec()) && d.hasSpec;
                f4 cur = ld_h4(HIST, x, y, sig * sb);
                // 5x5 local luma moments
                float m1 = 0.0f, m2 = 0.0f;
                for (int j = -2; j <= 2; j++)
                    for (int i = -2; i <= 2; i++) {
                        int px = x + i, py = y + j, gy = py + c.yOff;
                        float f = cur.x;
                        if (px >= 0 && px < c.W && gy >= 0 && gy < c.H && py >= 0 && py < c.resH) {
                            float zt = ld_f32(G, px, py, 0);
                            if (absf(zt) <= c.denoisingRange)
                                f = ld_h(HIST, px, py, sig * sb);
                        }
                        m1 += f;
                        m2 = fma_(f, f, m2);
                    }
                m1 *= 1.0f / 25.0f;
                m2 *= 1.0f / 25.0f;
                float sigma = sqrt_(fmax2(fma_(-m1, m1, m2), 0.0f));
                // stabilized luma history: surface motion footprint (validity bits from TA), virtual motion for specular
                auto fetchStab = [&](float pu, float pv, uint32_t bits, float& out) -> bool {
                    float px = fma_(pu, (float)c.Wprev, -0.5f), py = fma_(pv, (float)c.Hprev, -0.5f);
                    float fx0 = floorf(px), fy0 = floorf(py);
                    float fx = px - fx0, fy = py - fy0;
                    bool sane = fx0 >= -2.0f && fx0 <= (float)c.Wprev + 1.0f && fy0 >= -2.0f && fy0 <= (float)c.Hprev + 1.0f;
                    if (!sane)
                        return false;
                    int ix = (int)fx0, iy = (int)fy0;
                    float bw[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
                    float sum = 0.0f, wsum = 0.0f;
                    for (int i = 0; i < 4; i++)
                        if (bits & (1u << i)) {
                            sum = fma_(ld_h(STABP, ix + (i & 1), iy + (i >> 1) - c.yOff, sig * 2), bw[i], sum);
                            wsum += bw[i];
                        }
                    if (!(wsum > 0.0f))
                        return false;
                    out = sum * (rcp_(wsum));
                    return true;
                };
                float Yhist = cur.x;
                bool have = false;
                if (historyOk) {
                    float smbY = 0.0f;
                    bool smbOk = fetchStab(r.su, r.sv, data2 & 15u, smbY);
                    if (isSpec) {
                        float amount = (float)((data2 >> 8) & 255u) * (1.0f / 255.0f);
                        float vu, vv, vmbY = 0.0f;
                        bool vmbOk = amount > 0.0f && virtual_uv(c, r, ld_h(HT, x, y), g.roughness, g.mat, vu, vv) && fetchStab(vu, vv, (data2 >> 4) & 15u, vmbY);
                        if (smbOk && vmbOk) {
                            Yhist = lerpf(smbY, vmbY, amount);
                            have = true;
                        } else if (smbOk) {
                            Yhist = smbY;
                            have = true;
                        } else if (vmbOk) {
                            Yhist = vmbY;
                            have = true;
                        }
                    } else if (smbOk) {
                        Yhist = smbY;
                        have = true;
                    }
                }
                float Acur = A[isSpec ? 1 : 0];
                float Y = cur.x;
                float band = sigma * s.antilagSettings.luminanceSigmaScale;
                float dlt = fmax2(absf(Yhist - m1) - band, 0.0f) * rcps_(fmax2(Yhist, m1) + 1e-6f);
                float antilag = rcps_(fma_(dlt * s.antilagSettings.luminanceSensitivity, Acur, 1.0f));
                float Yclamped = clampf(Yhist, m1 - band, m1 + band);
                float stabFrames = have ? fmin2(Acur, maxStab) * antilag : 0.0f;
                float wHist = stabFrames * rcp_(1.0f + stabFrames);
                float Yout = lerpf(Y, Yclamped, wHist);
                float scale = (Yout + 1e-6f) * rcps_(Y + 1e-6f);
                f4 o = {Yout, cur.y * scale, cur.z * scale, cur.w};
                // returnHistoryLengthInsteadOfOcclusion (OCCLUSION variants): the single output channel reports the normalised history length
                if (d.occlusion && s.returnHistoryLengthInsteadOfOcclusion)
                    o.x = sat(Acur * (1.0f / fmax2((float)std::min<uint32_t>(s.maxAccumulatedFrameNum, 63), 1.0f)));
                st_h(STABC, x, y, Yout, sig * 2);
                if (dirOcc) {
                    f4 c1 = ld_h4(HIST, x, y, sig * sb + 8);
                    st_dir(*outP[sig], x, y, split ? dir_pass(*inP[sig], *in1P[sig], x, y) : f4{c1.x * scale, c1.y * scale, c1.z * scale, Yout});
                    continue;
                }
                store_signal(*outP[sig], x, y, split ? load_signal(*inP[sig], x, y, 0, occIn) : o, d.occlusion);
                if (d.sh) {
                    f4 c1 = ld_h4(HIST, x, y, sig * sb + 8);
                    st_h4(*out1P[sig], x, y, split ? ld_h4(*in1P[sig], x, y) : f4{c1.x * scale, c1.y * scale, c1.z * scale, c1.w});
                }
            }
        }
}


// --------------------------------------------------------------------------------------------------
// RELAX A-trous iteration (variance-guided 3x3 at stride 2^it). Iteration 0 reads the history written by HistoryFix
// (YCoCg + hitDist) and derives the variance from the accumulated luma moments (spatial 3x3 estimate while the history is
// short); later iterations ping-pong {YCoCg, variance} texels; the last one converts to linear RGB and writes OUT_*.
// Reference call sites: RelaxSettings fields Source/NRDSample.cpp:1642-1657 (atrousIterationNum, phi luminance, min luminance
// weight, depth threshold, lobe / roughness fraction, history threshold); outputs decoded by RELAX_BackEnd_UnpackRadiance
// Shaders/Composition.cs.hlsl:160-161.
// --------------------------------------------------------------------------------------------------
enum RelaxTrans { T_AT_A = T_NUM, T_AT_B };

void atrous(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1, int it, bool last) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const int sb = d.sh ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    (void)sb;
    const nrd::RelaxSettings& s = d.relax;
    const Plane& G = k.guide();
    const Plane& HIST = k.perm(P_HIST);
    const Plane& MOM = k.perm(P_STAB_A + k.cur);
    const Plane& D1 = k.perm(P_DATA1_A + k.cur);
    const Plane& IN = it == 0 ? HIST : k.trans(T_AT_A + ((it - 1) & 1));
    const Plane& OUTP = k.trans(T_AT_A + (it & 1));
    const Plane* outSlot[2] = {nullptr, nullptr};
    const PrepareMode pmA = prepare_mode(d); // split screen shows the noisy input (its dense copy when PrepareInputs ran)
    const Plane* inSlot[2] = {nullptr, nullptr};
    const Plane* out1Slot[2] = {nullptr, nullptr};
    const Plane* in1Slot[2] = {nullptr, nullptr};
    if (d.hasDiff) {
        outSlot[k.sigDiff()] = &k.slot(out_slot(d, false));
        inSlot[k.sigDiff()] = pmA.any ? &k.trans(T_PREP_D) : &k.slot(in_slot(d, false));
        out1Slot[k.sigDiff()] = &k.slot(out1_slot(false));
        in1Slot[k.sigDiff()] = pmA.sh1 ? &k.trans(T_PREP_D1) : &k.slot(in1_slot(false));
    }
    if (d.hasSpec) {
        outSlot[k.sigSpec()] = &k.slot(out_slot(d, true));
        inSlot[k.sigSpec()] = pmA.any ? &k.trans(T_PREP_S) : &k.slot(in_slot(d, true));
        out1Slot[k.sigSpec()] = &k.slot(out1_slot(true));
        in1Slot[k.sigSpec()] = pmA.sh1 ? &k.trans(T_PREP_S1) : &k.slot(in1_slot(true));
    }
    const int stride = 1 << it;
    const Plane& D2 = k.trans(T_DATA2);
    const bool relaxEdges = stride <= 4; // edge-stopping relaxation acts on the fine iterations only
    const float depthSens = fmax2(s.depthThreshold, 0.001f) * 4.0f;
    const float histThreshold = (float)s.spatialVarianceEstimationHistoryThreshold;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c.W; x++) {
            int gy0 = y + c.yOff;
            float u = ((float)x + 0.5f) * c.invW;
            bool split = last && u < c.splitScreen;
            Guide g = load_guide(G, x, y, c.denoisingRange);
            if (g.sky) {
                for (int sig = 0; sig < d.nsig; sig++) {
                    if (last) {
                        st_h4(*outSlot[sig], x, y, split ? ld_h4(*inSlot[sig], x, y) : f4{0, 0, 0, 0});
                        if (d.sh)
                            st_h4(*out1Slot[sig], x, y, split ? ld_h4(*in1Slot[sig], x, y) : f4{0, 0, 0, 0});
                    } else {
                        st_h4(OUTP, x, y, {0, 0, 0, 0}, sig * sb);
                        if (d.sh)
                            st_h4(OUTP, x, y, {0, 0, 0, 0}, sig * sb + 8);
                    }
                }
                continue;
            }
            PixelGeo pg = pixel_geo(c, g, x, gy0, depthSens);
            float A[2] = {0, 0};
            if (it == 0)
                unpack_data1(ld_u16(D1, x, y), A[0], A[1]);
            for (int sig = 0; sig < d.nsig; sig++) {
                bool isSpec = (sig == k.sigSpec()) && d.hasSpec;
                float rough = isSpec ? g.roughness : 1.0f;
                uint32_t minMat = isSpec ? s.minMaterialForSpecular : s.minMaterialForDiffuse;
                f4 c0 = ld_h4(IN, x, y, sig * sb);
                const float c0Y = signal_luma(c0, true); // luminance of the centre texel
                f4 sum1 = d.sh ? ld_h4(IN, x, y, sig * sb + 8) : f4{0, 0, 0, 0};
                float var;
                if (it == 0) {
                    float m2 = ld_h(MOM, x, y, sig * 2);
                    var = fmax2(fma_(-c0Y, c0Y, m2), 0.0f);
                    if (A[isSpec ? 1 : 0] < histThreshold) { // short history: 3x3 spatial estimate
                        float sy = 0.0f, sy2 = 0.0f, n = 0.0f;
                        for (int j = -1; j <= 1; j++)
                            for (int i = -1; i <= 1; i++) {
                                int px = x + i, py = y + j, gy = py + c.yOff;
                                if (px < 0 || px >= c.W || gy < 0 || gy >= c.H || py < 0 || py >= c.resH)
                                    continue;
                                if (!(absf(ld_f32(G, px, py, 0)) <= c.denoisingRange))
                                    continue;
                                float Y = ld_luma(HIST, px, py, sig * sb, true);
                                sy += Y;
                                sy2 = fma_(Y, Y, sy2);
                                n += 1.0f;
                            }
                        float inv = rcp_(n);
                        float my = sy * inv;
                        var = fmax2(var, fmax2(fma_(-my, my, sy2 * inv), 0.0f));
                    }
                    if (isSpec)
                        var = fma_(var, s.specularVarianceBoost, var);
                } else
                    var = c0.w;
                float sigma = wsqrt_(var);
                float phi = isSpec ? s.specularPhiLuminance : s.diffusePhiLuminance;
                float minLw = isSpec ? s.specularMinLuminanceWeight : s.diffuseMinLuminanceWeight;
                float invL = 0.3333f * wrcp_(fma_(phi, sigma, 1e-4f));
                float angle = spec_lobe_half_angle(rough) * s.lobeAngleFraction;
                if (isSpec)
                    angle += s.specularLobeAngleSlack * 0.017453292f; // degrees
                float normalW = wrcp_(fmax2(angle, NORMAL_ANGLE_MIN));
                normalW *= strand_normal_relax(c, g.mat, absf(g.z)); // CommonSettings::strandMaterialID: thin strands relax the normal test
                // confidenceDriven*: low history confidence (IN_*_CONFIDENCE) relaxes the luminance / normal edge stopping of both signals
                if (s.confidenceDrivenRelaxationMultiplier > 0.0f && c.confAvail) {
                    float conf = sample_confidence(k.slot(isSpec ? nrd::ResourceType::IN_SPEC_CONFIDENCE : nrd::ResourceType::IN_DIFF_CONFIDENCE), u, ((float)gy0 + 0.5f) * c.invH);
                    float cd = sat(s.confidenceDrivenRelaxationMultiplier * (1.0f - conf));
                    invL *= fma_(-cd, sat(s.confidenceDrivenLuminanceEdgeStoppingRelaxation), 1.0f);
                    normalW *= fma_(-cd, sat(s.confidenceDrivenNormalEdgeStoppingRelaxation), 1.0f);
                }
                // {luminance, normal, roughness}EdgeStoppingRelaxation (sample UI Source/NRDSample.cpp:1650): where the specular history
                // was reprojected with low confidence (DATA2 bits 16..23) the three edge-stopping terms are relaxed toward "accept"
                float roughRelax = 1.0f;
                if (isSpec && relaxEdges) {
                    float conf = (float)((ld_u32(D2, x, y) >> 16) & 255u) * (1.0f / 255.0f);
                    invL *= lerpf(1.0f, conf, sat(s.luminanceEdgeStoppingRelaxation));
                    normalW *= lerpf(1.0f, conf, sat(s.normalEdgeStoppingRelaxation));
                    roughRelax = lerpf(1.0f, conf, sat(s.roughnessEdgeStoppingRelaxation));
                }
                float normalW2 = nw_param(normalW);
                float roughA = wrcp_(lerpf(0.01f, 1.0f, sat(rough * s.roughnessFraction)));
                float roughB = -rough * roughA;
                f3 sum = {c0.x, c0.y, c0.z};
                float sumVar = var, wsum = 1.0f;
                for (int j = -1; j <= 1; j++)
                    for (int i = -1; i <= 1; i++) {
                        if (i == 0 && j == 0)
                            continue;
                        // a rejected tap (outside the frame / the held rows, sky, other material) enters with weight 0: its texel
                        // is fetched at the clamped position (always a finite value of an internal plane)
                        int px = x + i * stride, py = y + j * stride, gy = py + c.yOff;
                        bool valid = !(px < 0 || px >= c.W || gy < 0 || gy >= c.H || py < 0 || py >= c.resH);
                        const int loY = std::max(0, -c.yOff), hiY = std::min(c.resH, c.H - c.yOff) - 1;
                        int cpx = std::min(std::max(px, 0), c.W - 1), cpy = std::min(std::max(py, loY), hiY);
                        Guide gs = load_guide(G, cpx, cpy, c.denoisingRange);
                        valid = valid && !gs.sky && !material_mismatch(g.mat, gs.mat, minMat);
                        f4 sv = ld_h4(IN, cpx, cpy, sig * sb);
                        float vs = sv.w;
                        if (it == 0)
                            vs = fmax2(fma_(-signal_luma(sv, true), signal_luma(sv, true), ld_h(MOM, cpx, cpy, sig * 2)), 0.0f);
                        float w = 0.0f;
                        if (valid) {
                            w = (i == 0 || j == 0) ? 0.5f : 0.25f;
                            w *= geo_weight(pg, (float)px, (float)gy, gs.z);
                            w *= normal_weight(normal_dist2(g.nw, gs.nw), normalW2);
                            if (isSpec && s.enableRoughnessEdgeStopping) {
                                float rw = smoothstep01(1.0f - absf(fma_(gs.roughness, roughA, roughB)));
                                w *= relaxEdges ? lerpf(1.0f, rw, roughRelax) : rw;
                            }
                            w *= fmax2(exp_weight(absf(signal_luma(sv, true) - c0Y) * invL), minLw);
                        }
                        sum = {fma_(sv.x, w, sum.x), fma_(sv.y, w, sum.y), fma_(sv.z, w, sum.z)};
                        if (d.sh)
                            sum1 = fma4(ld_h4(IN, cpx, cpy, sig * sb + 8), w, sum1);
                        sumVar = fma_(vs, w * w, sumVar);
                        wsum += w;
                    }
                float inv = wrcp_(wsum);
                f3 o = mul3(sum, inv);
                float ov = sumVar * inv * inv;
                if (last) {
                    f3 rgb = RELAX_LINEAR_RGB ? f3{fmax2(o.x, 0.0f), fmax2(o.y, 0.0f), fmax2(o.z, 0.0f)} : ycocg_to_linear(o);
                    float hitDist = ld_h(HIST, x, y, sig * sb + 6);
                    st_h4(*outSlot[sig], x, y, split ? ld_h4(*inSlot[sig], x, y) : f4{rgb.x, rgb.y, rgb.z, hitDist});
                    if (d.sh)
                        st_h4(*out1Slot[sig], x, y, split ? ld_h4(*in1Slot[sig], x, y) : mul4(sum1, inv));
                } else {
                    st_h4(OUTP, x, y, {o.x, o.y, o.z, ov}, sig * sb);
                    if (d.sh)
                        st_h4(OUTP, x, y, mul4(sum1, inv), sig * sb + 8);
                }
            }
        }
}

} // namespace

// --------------------------------------------------------------------------------------------------
// Validation overlay (CommonSettings::enableValidation, OUT_VALIDATION bound at Source/NRDSample.cpp:452, RGBA8): per pixel
// {diffuse accumulated frames / 63, specular accumulated frames / 63, |viewZ| / denoisingRange, virtual-motion amount}; 0 on sky
// --------------------------------------------------------------------------------------------------
void validation(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const Plane& G = k.guide();
    const Plane& D1 = k.perm(P_DATA1_A + k.cur);
    const Plane& D2 = k.trans(T_DATA2);
    const Plane& OUT = k.slot(nrd::ResourceType::OUT_VALIDATION);
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c.W; x++) {
            Guide g = load_guide(G, x, y, c.denoisingRange);
            uint32_t packed = 0;
            if (!g.sky) {
                float dA, sA;
                unpack_data1(ld_u16(D1, x, y), dA, sA);
                uint32_t r = (uint32_t)floorf(fma_(sat(dA * (1.0f / 63.0f)), 255.0f, 0.5f));
                uint32_t gg = (uint32_t)floorf(fma_(sat(sA * (1.0f / 63.0f)), 255.0f, 0.5f));
                uint32_t b = (uint32_t)floorf(fma_(sat(absf(g.z) * rcp_(c.denoisingRange)), 255.0f, 0.5f));
                uint32_t a = (ld_u32(D2, x, y) >> 8) & 255u;
                packed = r | (gg << 8) | (b << 16) | (a << 24);
            }
            st_u32(OUT, x, y, packed);
        }
}
static void push_validation_pass(Instance& I, DenoiserState& d, const char* name, uint32_t guide, uint32_t data1, uint32_t data2) {
    if (!I.common.enableValidation || !I.slots[(size_t)nrd::ResourceType::OUT_VALIDATION].p)
        return;
    Pass p;
    p.name = name;
    p.kernel = "nrd_reblur_validation";
    p.haloRows = 0;
    p.bytesPerPixel = (float)GUIDE_BYTES + 2.0f + 4.0f + 4.0f;
    p.read = {guide, data1, data2};
    p.written = {enc_slot(nrd::ResourceType::OUT_VALIDATION)};
    p.run = validation;
    d.passes.push_back(p);
}

void reblur_describe(DenoiserState& d, std::vector<PoolPlane>& perm, std::vector<PoolPlane>& trans) {
    uint32_t fmtRad = (uint32_t)(d.nsig == 2 ? nrd::Format::RGBA32_UINT : nrd::Format::RGBA16_SFLOAT);
    uint32_t fmtLum = (uint32_t)(d.nsig == 2 ? nrd::Format::RG16_SFLOAT : nrd::Format::R16_SFLOAT);
    uint32_t bRad = 8u * d.nsig * (d.sh ? 2u : 1u), bLum = 2u * d.nsig; // SH mode: SH0 + SH1 texels per signal
    perm.push_back({"REBLUR::Guide_A", (uint32_t)nrd::Format::RG32_UINT, 8, 1});
    perm.push_back({"REBLUR::Guide_B", (uint32_t)nrd::Format::RG32_UINT, 8, 1});
    perm.push_back({"REBLUR::Data1_A", (uint32_t)nrd::Format::R16_UINT, 2, 1});
    perm.push_back({"REBLUR::Data1_B", (uint32_t)nrd::Format::R16_UINT, 2, 1});
    perm.push_back({"REBLUR::History", fmtRad, bRad, 1});
    perm.push_back({"REBLUR::FastHistory_A", fmtLum, bLum, 1});
    perm.push_back({"REBLUR::FastHistory_B", fmtLum, bLum, 1});
    perm.push_back({"REBLUR::StabilizedLuma_A", fmtLum, bLum, 1});
    perm.push_back({"REBLUR::StabilizedLuma_B", fmtLum, bLum, 1});
    trans.push_back({"REBLUR::Tiles", (uint32_t)nrd::Format::R8_UINT, 1, 16});
    trans.push_back({"REBLUR::Tmp1", fmtRad, bRad, 1});
    trans.push_back({"REBLUR::Tmp2", fmtRad, bRad, 1});
    trans.push_back({"REBLUR::Data1_Tmp", (uint32_t)nrd::Format::R16_UINT, 2, 1});
    trans.push_back({"REBLUR::Data2", (uint32_t)nrd::Format::R32_UINT, 4, 1});
    trans.push_back({"REBLUR::SpecHitDistForTracking", (uint32_t)nrd::Format::R16_SFLOAT, 2, 1});
    // PrepareInputs outputs (checkerboard resolve / hit distance reconstruction): dense RGBA16F copies of the noisy inputs;
    // the SH1 copies are only full-size in SH mode
    trans.push_back({"REBLUR::Prepared_Diff", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, 1});
    trans.push_back({"REBLUR::Prepared_Spec", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, 1});
    trans.push_back({"REBLUR::Prepared_DiffSh1", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, (uint16_t)(d.sh ? 1 : 16)});
    trans.push_back({"REBLUR::Prepared_SpecSh1", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, (uint16_t)(d.sh ? 1 : 16)});
    // tap texels of Blur / PostBlur (TapTexel; radiance flavours only): _A HistoryFix -> Blur, _B Blur -> PostBlur
    const bool tap = tap_texels(d);
    trans.push_back({"REBLUR::Tap_Diff_A", (uint32_t)nrd::Format::RGBA32_UINT, 16, (uint16_t)(tap && d.hasDiff ? 1 : 16)});
    trans.push_back({"REBLUR::Tap_Spec_A", (uint32_t)nrd::Format::RGBA32_UINT, 16, (uint16_t)(tap && d.hasSpec ? 1 : 16)});
    trans.push_back({"REBLUR::Tap_Diff_B", (uint32_t)nrd::Format::RGBA32_UINT, 16, (uint16_t)(tap && d.hasDiff ? 1 : 16)});
    trans.push_back({"REBLUR::Tap_Spec_B", (uint32_t)nrd::Format::RGBA32_UINT, 16, (uint16_t)(tap && d.hasSpec ? 1 : 16)});
}

// the tap planes of the signals present, diffuse first (base = T_TAP_D_A or T_TAP_D_B)
static void push_tap_planes(const DenoiserState& d, uint32_t tb, int base, std::vector<uint32_t>& list) {
    if (d.hasDiff)
        list.push_back(enc_trans(tb + base));
    if (d.hasSpec)
        list.push_back(enc_trans(tb + base + 1));
}

void reblur_build(Instance& I, DenoiserState& d) {
    using RT = nrd::ResourceType;
    int cur = (int)(d.frameCounter & 1);
    uint32_t pb = d.permBase, tb = d.transBase;
    auto P = [&](int i) { return enc_perm(pb + i); };
    auto T = [&](int i) { return enc_trans(tb + i); };
    float n = (float)d.nsig;
    float nr = n * (d.sh ? 2.0f : 1.0f); // radiance texels per pixel (SH mode doubles them)
    const nrd::ReblurSettings& s = d.reblur;
    ReblurReach rr = reblur_reach(s);
    const float GB = (float)GUIDE_BYTES; // guide texel bytes
    const bool tap = tap_texels(d);
    {
        Pass p;
        p.name = "REBLUR::ClassifyTiles";
        p.kernel = "nrd_reblur_classify_tiles";
        p.haloRows = 0;
        p.bytesPerPixel = 4 + 4 + GB + 1.0f / 256.0f;
        p.read = {enc_slot(RT::IN_VIEWZ), enc_slot(RT::IN_NORMAL_ROUGHNESS)};
        p.written = {P(P_GUIDE_A + cur), T(T_TILES)};
        p.tileGrid = true;
        p.allRows = true;
        p.run = classify_tiles;
        d.passes.push_back(p);
    }
    const PrepareMode pm = prepare_mode(d);
    if (pm.any) {
        Pass p;
        p.name = "REBLUR::PrepareInputs";
        p.kernel = "nrd_reblur_prepare_inputs";
        p.haloRows = (uint16_t)pm.radius;
        float inB = (d.occlusion ? 2.0f : 8.0f) * (pm.checker ? 0.5f : 1.0f);
        p.bytesPerPixel = GB + n * (inB + 8.0f) + (pm.sh1 ? n * ((d.dirOcc ? 0.0f : 4.0f) + 8.0f) : 0.0f);
        p.read = {P(P_GUIDE_A + cur)};
        for (int si = 0; si < 2; si++) {
            if (!(si ? d.hasSpec : d.hasDiff))
                continue;
            p.read.push_back(enc_slot(in_slot(d, si != 0)));
            p.written.push_back(T(T_PREP_D + si));
            if (pm.sh1) {
                if (!d.dirOcc)
                    p.read.push_back(enc_slot(in1_slot(si != 0)));
                p.written.push_back(T(T_PREP_D1 + si));
            }
        }
        p.run = prepare_inputs;
        d.passes.push_back(p);
    }
    // the PrePass of rows [y0, y1) into `result` (Tmp1, or the scratch rows of the fused dispatch)
    static const auto prepass_rows = [](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1, const Plane* result) {
        Ctx k{I, d, c, (int)(d.frameCounter & 1)};
        const int sb = d.sh ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
        SpatialIO io = {};
        io.reach = reblur_reach(d.reblur).pre;
        for (int sig = 0; sig < d.nsig; sig++) {
            bool isSpec = (sig == k.sigSpec()) && d.hasSpec;
            PrepareMode pm = prepare_mode(d);
            io.in[sig] = pm.any ? &k.trans(T_PREP_D + (isSpec ? 1 : 0)) : &k.slot(in_slot(d, isSpec));
            io.in1[sig] = d.sh ? (pm.sh1 ? &k.trans(T_PREP_D1 + (isSpec ? 1 : 0)) : &k.slot(in1_slot(isSpec))) : nullptr;
            io.inOff[sig] = 0;
            io.out[sig] = result;
            io.outOff[sig] = sig * sb;
        }
        spatial_filter(k, PRE, io, y0, y1);
    };
    const bool fused = fused_prepass(I, d);
    std::vector<uint32_t> prepassInputs;
    for (int si = 0; si < 2; si++) {
        if (!(si ? d.hasSpec : d.hasDiff))
            continue;
        prepassInputs.push_back(pm.any ? T(T_PREP_D + si) : enc_slot(in_slot(d, si != 0)));
        if (d.sh)
            prepassInputs.push_back(pm.sh1 ? T(T_PREP_D1 + si) : enc_slot(in1_slot(si != 0)));
    }
    if (fused) {
        // PrePass + TemporalAccumulation as ONE dispatch (csrc/nrd_reblur.hip spatial_pixel<..., FUSED>): TemporalAccumulation reads the
        // PrePass result at its own pixel only, so the kernel keeps it in registers and Tmp1 is neither written nor read. Here: the
        // PrePass of the rows goes to private scratch rows with Tmp1's texel layout, TemporalAccumulation reads those - the same
        // roundings (packed fp16 texels) as through the plane
        Pass p;
        p.name = "REBLUR::PrePassTemporalAccumulation";
        p.kernel = "nrd_reblur_prepass_temporal_accumulation";
        p.haloRows = (uint16_t)rr.pre;
        p.bytesPerPixel = GB + 8 * nr + 8 + GB + 2 + 8 * nr + 2 * n + 8 * nr + 2 * n + 2 + 4 + (d.hasSpec ? 2 : 0);
        p.read = {P(P_GUIDE_A + cur)};
        p.read.insert(p.read.end(), prepassInputs.begin(), prepassInputs.end());
        for (uint32_t r : {P(P_GUIDE_A + (cur ^ 1)), enc_slot(RT::IN_MV), P(P_HIST), P(P_FAST_A + (cur ^ 1)), P(P_DATA1_A + (cur ^ 1))})
            p.read.push_back(r);
        if (I.common.isDisocclusionThresholdMixAvailable)
            p.read.push_back(enc_slot(RT::IN_DISOCCLUSION_THRESHOLD_MIX));
        p.written = {T(T_HITTRACK), T(T_TMP2), P(P_FAST_A + cur), T(T_DATA1), T(T_DATA2)};
        p.reprojected = {P(P_GUIDE_A + (cur ^ 1)), P(P_HIST), P(P_FAST_A + (cur ^ 1)), P(P_DATA1_A + (cur ^ 1))};
        p.run = [](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
            if (y1 <= y0)
                return;
            Ctx k{I, d, c, (int)(d.frameCounter & 1)};
            Plane scratch = k.trans(T_TMP1); // same width, texel size and format; own rows
            scratch.pitch = (uint32_t)scratch.w * scratch.bpt;
            std::vector<uint8_t> rows((size_t)(y1 - y0) * scratch.pitch, 0);
            scratch.p = rows.data() - (size_t)y0 * scratch.pitch;
            prepass_rows(I, d, c, y0, y1, &scratch);
            temporal_accumulation_rows(I, d, c, y0, y1, &scratch);
        };
        d.passes.push_back(p);
    } else {
        Pass p;
        p.name = "REBLUR::PrePass";
        p.kernel = "nrd_reblur_prepass";
        p.haloRows = (uint16_t)rr.pre;
        p.bytesPerPixel = GB + 8 * nr + 8 * nr + (d.hasSpec ? 2 : 0);
        p.read = {P(P_GUIDE_A + cur)};
        p.read.insert(p.read.end(), prepassInputs.begin(), prepassInputs.end());
        p.written = {T(T_TMP1), T(T_HITTRACK)};
        p.run = [](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) { prepass_rows(I, d, c, y0, y1, &I.trans[d.transBase + T_TMP1]); };
        d.passes.push_back(p);
    }
    if (!fused) {
        Pass p;
        p.name = "REBLUR::TemporalAccumulation";
        p.kernel = "nrd_reblur_temporal_accumulation";
        p.haloRows = 0; // previous-frame planes are read at motion-displaced rows: the tiler adds its motion margin
        p.bytesPerPixel = GB + 8 + GB + 2 + 8 * nr + 8 * nr + 2 * n + (d.hasSpec ? 2 : 0) + 8 * nr + 2 * n + 2 + 4;
        p.read = {P(P_GUIDE_A + cur), P(P_GUIDE_A + (cur ^ 1)), enc_slot(RT::IN_MV), T(T_TMP1), P(P_HIST), P(P_FAST_A + (cur ^ 1)), P(P_DATA1_A + (cur ^ 1)), T(T_HITTRACK)};
        if (I.common.isDisocclusionThresholdMixAvailable)
            p.read.push_back(enc_slot(RT::IN_DISOCCLUSION_THRESHOLD_MIX));
        p.written = {T(T_TMP2), P(P_FAST_A + cur), T(T_DATA1), T(T_DATA2)};
        p.reprojected = {P(P_GUIDE_A + (cur ^ 1)), P(P_HIST), P(P_FAST_A + (cur ^ 1)), P(P_DATA1_A + (cur ^ 1))};
        p.run = temporal_accumulation;
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "REBLUR::HistoryFix";
        p.kernel = "nrd_reblur_history_fix";
        p.haloRows = (uint16_t)(2 * s.historyFixBasePixelStride + 2);
        p.bytesPerPixel = GB + 2 + 8 * nr + 2 * n + (tap ? 16 * n : 8 * nr) + 2;
        p.read = {P(P_GUIDE_A + cur), T(T_TMP2), T(T_DATA1), P(P_FAST_A + cur)};
        p.reach = {{P(P_FAST_A + cur), (uint16_t)2}}; // the 5x5 clamping window; the reconstruction taps read guide, signal and speeds
        if (tap) {
            push_tap_planes(d, tb, T_TAP_D_A, p.written);
            for (uint32_t code : p.written) // the tap texels start with the pixel's guide texel as it is
                p.prefix.push_back({code, P(P_GUIDE_A + cur)});
            p.written.push_back(P(P_DATA1_A + cur));
        } else
            p.written = {T(T_TMP1), P(P_DATA1_A + cur)};
        p.run = history_fix;
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "REBLUR::Blur";
        p.kernel = "nrd_reblur_blur";
        p.haloRows = (uint16_t)rr.blur;
        p.own = {P(P_DATA1_A + cur)};
        if (tap) { // the tap texels carry the guide: no guide plane access
            p.bytesPerPixel = 2 + 16 * n + 16 * n;
            p.read = {P(P_DATA1_A + cur)};
            push_tap_planes(d, tb, T_TAP_D_A, p.read);
            push_tap_planes(d, tb, T_TAP_D_B, p.written);
            for (uint32_t code : p.written) // Blur hands the guide part of its input texel on
                p.prefix.push_back({code, P(P_GUIDE_A + cur)});
        } else {
            p.bytesPerPixel = GB + 2 + 8 * nr + 8 * nr;
            p.read = {P(P_GUIDE_A + cur), P(P_DATA1_A + cur), T(T_TMP1)};
            p.written = {T(T_TMP2)};
        }
        p.run = [](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
            Ctx k{I, d, c, (int)(d.frameCounter & 1)};
            const int sb = d.sh ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
            (void)sb;
            SpatialIO io = {};
            io.reach = reblur_reach(d.reblur).blur;
            for (int sig = 0; sig < d.nsig; sig++) {
                const int spec = (sig == k.sigSpec() && d.hasSpec) ? 1 : 0;
                io.in[sig] = tap_texels(d) ? &k.trans(T_TAP_D_A + spec) : &k.trans(T_TMP1);
                io.out[sig] = tap_texels(d) ? &k.trans(T_TAP_D_B + spec) : &k.trans(T_TMP2);
                io.inOff[sig] = io.outOff[sig] = sig * sb;
            }
            spatial_filter(k, BLUR, io, y0, y1);
        };
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "REBLUR::PostBlur";
        p.kernel = "nrd_reblur_post_blur";
        p.haloRows = (uint16_t)rr.post;
        p.own = {P(P_DATA1_A + cur)};
        if (tap) {
            p.bytesPerPixel = 2 + 16 * n + 8 * nr;
            p.read = {P(P_DATA1_A + cur)};
            push_tap_planes(d, tb, T_TAP_D_B, p.read);
        } else {
            p.bytesPerPixel = GB + 2 + 8 * nr + 8 * nr;
            p.read = {P(P_GUIDE_A + cur), P(P_DATA1_A + cur), T(T_TMP2)};
        }
        p.written = {P(P_HIST)};
        p.run = [](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
            Ctx k{I, d, c, (int)(d.frameCounter & 1)};
            const int sb = d.sh ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
            (void)sb;
            SpatialIO io = {};
            io.reach = reblur_reach(d.reblur).post;
            for (int sig = 0; sig < d.nsig; sig++) {
                const int spec = (sig == k.sigSpec() && d.hasSpec) ? 1 : 0;
                io.in[sig] = tap_texels(d) ? &k.trans(T_TAP_D_B + spec) : &k.trans(T_TMP2);
                io.out[sig] = &k.perm(P_HIST);
                io.inOff[sig] = io.outOff[sig] = sig * sb;
            }
            spatial_filter(k, POST, io, y0, y1);
        };
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "REBLUR::TemporalStabilization";
        p.kernel = "nrd_reblur_temporal_stabilization";
        p.haloRows = 2;
        p.own = {P(P_DATA1_A + cur), T(T_DATA2), T(T_HITTRACK)}; // (guide and history: the 5x5 window, = the pass's halo)
        p.reprojected = {P(P_STAB_A + (cur ^ 1))};
        p.bytesPerPixel = GB + 2 + 4 + 8 + 8 * nr + 2 * n + (d.hasSpec ? 2 : 0) + 8 * nr + 2 * n;
        p.read = {P(P_GUIDE_A + cur), P(P_DATA1_A + cur), T(T_DATA2), enc_slot(RT::IN_MV), P(P_HIST), P(P_STAB_A + (cur ^ 1)), T(T_HITTRACK)};
        p.written = {P(P_STAB_A + cur)};
        if (d.hasDiff) {
            p.written.push_back(enc_slot(out_slot(d, false)));
            if (d.sh && !d.dirOcc)
                p.written.push_back(enc_slot(out1_slot(false)));
            p.read.push_back(enc_slot(in_slot(d, false)));
            if (d.sh && !d.dirOcc)
                p.read.push_back(enc_slot(in1_slot(false)));
        }
        if (d.hasSpec) {
            p.written.push_back(enc_slot(out_slot(d, true)));
            if (d.sh)
                p.written.push_back(enc_slot(out1_slot(true)));
            p.read.push_back(enc_slot(in_slot(d, true)));
            if (d.sh)
                p.read.push_back(enc_slot(in1_slot(true)));
        }
        p.run = temporal_stabilization;
        d.passes.push_back(p);
    }
    push_validation_pass(I, d, "REBLUR::Validation", P(P_GUIDE_A + cur), P(P_DATA1_A + cur), T(T_DATA2));
}


// ---- RELAX = shared front half (ClassifyTiles, PrePass, TemporalAccumulation, HistoryFix) + A-trous iterations --------
static nrd::ReblurSettings relax_as_reblur(const nrd::RelaxSettings& r) {
    nrd::ReblurSettings s = {};
    s.hitDistanceParameters = {1.0f, 0.0f, 1.0f, 0.0f}; // hit distances stay in world units
    s.maxAccumulatedFrameNum = r.diffuseMaxAccumulatedFrameNum;
    s.maxFastAccumulatedFrameNum = r.diffuseMaxFastAccumulatedFrameNum;
    s.historyFixFrameNum = r.historyFixFrameNum;
    s.historyFixBasePixelStride = r.historyFixBasePixelStride;
    s.diffusePrepassBlurRadius = r.diffusePrepassBlurRadius;
    s.specularPrepassBlurRadius = r.specularPrepassBlurRadius;
    s.minHitDistanceWeight = r.minHitDistanceWeight;
    s.lobeAngleFraction = r.lobeAngleFraction;
    s.roughnessFraction = r.roughnessFraction;
    s.fastHistoryClampingSigmaScale = r.fastHistoryClampingSigmaScale;
    s.minMaterialForDiffuse = r.minMaterialForDiffuse;
    s.minMaterialForSpecular = r.minMaterialForSpecular;
    s.checkerboardMode = r.checkerboardMode;
    s.hitDistanceReconstructionMode = r.hitDistanceReconstructionMode;
    s.enableAntiFirefly = r.enableAntiFirefly;
    return s;
}

void relax_describe(DenoiserState& d, std::vector<PoolPlane>& perm, std::vector<PoolPlane>& trans) {
    uint32_t fmtRad = (uint32_t)(d.nsig == 2 ? nrd::Format::RGBA32_UINT : nrd::Format::RGBA16_SFLOAT);
    uint32_t fmtLum = (uint32_t)(d.nsig == 2 ? nrd::Format::RG16_SFLOAT : nrd::Format::R16_SFLOAT);
    uint32_t bRad = 8u * d.nsig * (d.sh ? 2u : 1u), bLum = 2u * d.nsig;
    perm.push_back({"RELAX::Guide_A", (uint32_t)nrd::Format::RG32_UINT, 8, 1});
    perm.push_back({"RELAX::Guide_B", (uint32_t)nrd::Format::RG32_UINT, 8, 1});
    perm.push_back({"RELAX::HistoryLength_A", (uint32_t)nrd::Format::R16_UINT, 2, 1});
    perm.push_back({"RELAX::HistoryLength_B", (uint32_t)nrd::Format::R16_UINT, 2, 1});
    perm.push_back({"RELAX::History", fmtRad, bRad, 1});
    perm.push_back({"RELAX::FastHistory_A", fmtLum, bLum, 1});
    perm.push_back({"RELAX::FastHistory_B", fmtLum, bLum, 1});
    perm.push_back({"RELAX::Moments_A", fmtLum, bLum, 1});
    perm.push_back({"RELAX::Moments_B", fmtLum, bLum, 1});
    trans.push_back({"RELAX::Tiles", (uint32_t)nrd::Format::R8_UINT, 1, 16});
    trans.push_back({"RELAX::Tmp1", fmtRad, bRad, 1});
    trans.push_back({"RELAX::Tmp2", fmtRad, bRad, 1});
    trans.push_back({"RELAX::HistoryLength_Tmp", (uint32_t)nrd::Format::R16_UINT, 2, 1});
    trans.push_back({"RELAX::Data2", (uint32_t)nrd::Format::R32_UINT, 4, 1});
    trans.push_back({"RELAX::SpecHitDistForTracking", (uint32_t)nrd::Format::R16_SFLOAT, 2, 1});
    // PrepareInputs outputs (checkerboard resolve / hit distance reconstruction): dense RGBA16F copies of the noisy inputs;
    // the SH1 copies are only full-size in SH mode
    trans.push_back({"RELAX::Prepared_Diff", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, 1});
    trans.push_back({"RELAX::Prepared_Spec", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, 1});
    trans.push_back({"RELAX::Prepared_DiffSh1", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, (uint16_t)(d.sh ? 1 : 16)});
    trans.push_back({"RELAX::Prepared_SpecSh1", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, (uint16_t)(d.sh ? 1 : 16)});
    trans.push_back({"RELAX::Atrous_A", fmtRad, bRad, 1});
    trans.push_back({"RELAX::Atrous_B", fmtRad, bRad, 1});
}

void relax_build(Instance& I, DenoiserState& d) {
    using RT = nrd::ResourceType;
    d.reblur = relax_as_reblur(d.relax); // the shared passes read their parameters from here
    int cur = (int)(d.frameCounter & 1);
    uint32_t pb = d.permBase, tb = d.transBase;
    auto P = [&](int i) { return enc_perm(pb + i); };
    auto T = [&](int i) { return enc_trans(tb + i); };
    float n = (float)d.nsig;
    float nr = n * (d.sh ? 2.0f : 1.0f); // radiance texels per pixel (SH mode doubles them)
    const nrd::ReblurSettings& s = d.reblur;
    ReblurReach rr = reblur_reach(s);
    const float GB = (float)GUIDE_BYTES;
    float sp = d.hasSpec ? 2.0f : 0.0f;
    {
        Pass p;
        p.name = "RELAX::ClassifyTiles";
        p.kernel = "nrd_reblur_classify_tiles";
        p.haloRows = 0;
        p.bytesPerPixel = 4 + 4 + GB + 1.0f / 256.0f;
        p.read = {enc_slot(RT::IN_VIEWZ), enc_slot(RT::IN_NORMAL_ROUGHNESS)};
        p.written = {P(P_GUIDE_A + cur), T(T_TILES)};
        p.tileGrid = true;
        p.allRows = true;
        p.run = classify_tiles;
        d.passes.push_back(p);
    }
    const PrepareMode pm = prepare_mode(d);
    if (pm.any) {
        Pass p;
        p.name = "RELAX::PrepareInputs";
        p.kernel = "nrd_reblur_prepare_inputs";
        p.haloRows = (uint16_t)pm.radius;
        float inB = (d.occlusion ? 2.0f : 8.0f) * (pm.checker ? 0.5f : 1.0f);
        p.bytesPerPixel = GB + n * (inB + 8.0f) + (pm.sh1 ? n * ((d.dirOcc ? 0.0f : 4.0f) + 8.0f) : 0.0f);
        p.read = {P(P_GUIDE_A + cur)};
        for (int si = 0; si < 2; si++) {
            if (!(si ? d.hasSpec : d.hasDiff))
                continue;
            p.read.push_back(enc_slot(in_slot(d, si != 0)));
            p.written.push_back(T(T_PREP_D + si));
            if (pm.sh1) {
                if (!d.dirOcc)
                    p.read.push_back(enc_slot(in1_slot(si != 0)));
                p.written.push_back(T(T_PREP_D1 + si));
            }
        }
        p.run = prepare_inputs;
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "RELAX::PrePass";
        p.kernel = "nrd_reblur_prepass";
        p.haloRows = (uint16_t)rr.pre;
        p.bytesPerPixel = GB + 8 * nr + 8 * nr + sp;
        p.read = {P(P_GUIDE_A + cur)};
        for (int si = 0; si < 2; si++) {
            if (!(si ? d.hasSpec : d.hasDiff))
                continue;
            p.read.push_back(pm.any ? T(T_PREP_D + si) : enc_slot(in_slot(d, si != 0)));
            if (d.sh)
                p.read.push_back(pm.sh1 ? T(T_PREP_D1 + si) : enc_slot(in1_slot(si != 0)));
        }
        p.written = {T(T_TMP1), T(T_HITTRACK)};
        p.run = [](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
            Ctx k{I, d, c, (int)(d.frameCounter & 1)};
            const int sb = d.sh ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
            (void)sb;
            SpatialIO io = {};
            io.reach = reblur_reach(d.reblur).pre;
            for (int sig = 0; sig < d.nsig; sig++) {
                bool isSpec = (sig == k.sigSpec()) && d.hasSpec;
                PrepareMode pm = prepare_mode(d);
                io.in[sig] = pm.any ? &k.trans(T_PREP_D + (isSpec ? 1 : 0)) : &k.slot(in_slot(d, isSpec));
                io.in1[sig] = d.sh ? (pm.sh1 ? &k.trans(T_PREP_D1 + (isSpec ? 1 : 0)) : &k.slot(in1_slot(isSpec))) : nullptr;
                io.inOff[sig] = 0;
                io.out[sig] = &k.trans(T_TMP1);
                io.outOff[sig] = sig * sb;
            }
            spatial_filter(k, PRE, io, y0, y1);
        };
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "RELAX::TemporalAccumulation";
        p.kernel = "nrd_reblur_temporal_accumulation";
        p.haloRows = 0;
        p.bytesPerPixel = GB + 8 + GB + 2 + 8 * nr + 8 * nr + 2 * n + 2 * n + sp + 8 * nr + 2 * n + 2 * n + 2 + 4;
        p.read = {P(P_GUIDE_A + cur), P(P_GUIDE_A + (cur ^ 1)), enc_slot(RT::IN_MV), T(T_TMP1), P(P_HIST), P(P_FAST_A + (cur ^ 1)),
                  P(P_DATA1_A + (cur ^ 1)), P(P_STAB_A + (cur ^ 1)), T(T_HITTRACK)};
        if (I.common.isDisocclusionThresholdMixAvailable)
            p.read.push_back(enc_slot(RT::IN_DISOCCLUSION_THRESHOLD_MIX));
        p.written = {T(T_TMP2), P(P_FAST_A + cur), P(P_STAB_A + cur), T(T_DATA1), T(T_DATA2)};
        p.reprojected = {P(P_GUIDE_A + (cur ^ 1)), P(P_HIST), P(P_FAST_A + (cur ^ 1)), P(P_DATA1_A + (cur ^ 1)), P(P_STAB_A + (cur ^ 1))};
        p.run = temporal_accumulation;
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "RELAX::HistoryFix";
        p.kernel = "nrd_reblur_history_fix";
        p.haloRows = (uint16_t)(2 * s.historyFixBasePixelStride + 2);
        p.bytesPerPixel = GB + 2 + 8 * nr + 2 * n + 2 * n + 8 * nr + 2;
        p.read = {P(P_GUIDE_A + cur), T(T_TMP2), T(T_DATA1), P(P_FAST_A + cur), P(P_STAB_A + cur)};
        p.reach = {{P(P_FAST_A + cur), (uint16_t)2}};
        p.own = {P(P_STAB_A + cur)};
        p.written = {P(P_HIST), P(P_DATA1_A + cur)};
        p.run = history_fix;
        d.passes.push_back(p);
    }
    int iters = (int)std::min<uint32_t>(std::max<uint32_t>(d.relax.atrousIterationNum, 2), 8);
    for (int it = 0; it < iters; it++) {
        bool last = it == iters - 1;
        Pass p;
        static const char* atrousNames[8] = {"RELAX::Atrous0", "RELAX::Atrous1", "RELAX::Atrous2", "RELAX::Atrous3", "RELAX::Atrous4", "RELAX::Atrous5", "RELAX::Atrous6", "RELAX::Atrous7"};
        p.name = atrousNames[it];
        p.kernel = "nrd_relax_atrous";
        p.haloRows = (uint16_t)(1 << it);
        const bool relaxEdges = it <= 2 && d.hasSpec; // reads the reprojection confidence (DATA2)
        p.bytesPerPixel = GB + (it == 0 ? 2 + 8 * nr + 2 * n : 8 * nr) + (last ? 8 * nr : 0.0f) + 8 * nr + (relaxEdges ? 4.0f : 0.0f);
        p.read = {P(P_GUIDE_A + cur)};
        if (relaxEdges)
            p.read.push_back(T(T_DATA2));
        if (it == 0) {
            p.read.push_back(P(P_DATA1_A + cur));
            p.read.push_back(P(P_HIST));
            p.read.push_back(P(P_STAB_A + cur));
        } else
            p.read.push_back(T(T_AT_A + ((it - 1) & 1)));
        if (last) {
            p.read.push_back(P(P_HIST));
            if (d.hasDiff) {
                p.written.push_back(enc_slot(out_slot(d, false)));
                p.read.push_back(enc_slot(in_slot(d, false)));
                if (d.sh) {
                    p.written.push_back(enc_slot(out1_slot(false)));
                    p.read.push_back(enc_slot(in1_slot(false)));
                }
            }
            if (d.hasSpec) {
                p.written.push_back(enc_slot(out_slot(d, true)));
                p.read.push_back(enc_slot(in_slot(d, true)));
                if (d.sh) {
                    p.written.push_back(enc_slot(out1_slot(true)));
                    p.read.push_back(enc_slot(in1_slot(true)));
                }
            }
        } else
            p.written = {T(T_AT_A + (it & 1))};
        p.run = [it, last](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) { atrous(I, d, c, y0, y1, it, last); };
        d.passes.push_back(p);
    }
    push_validation_pass(I, d, "RELAX::Validation", P(P_GUIDE_A + cur), P(P_DATA1_A + cur), T(T_DATA2));
}

} // namespace orc
