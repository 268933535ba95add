// orc_core.h - CPU ORACLE internals (test infrastructure, NOT product code; parity unpinned vs upstream NRD).
// Instance bookkeeping, plane access, per-frame constants. See oracle/README.md.
#pragma once

#include "../include/NRDDescs.h"
#include "../include/NRDSettings.h"
#include "orc_math.h"

#include <functional>
#include <string>
#include <vector>

namespace orc {

struct Plane {
    uint8_t* p = nullptr;
    uint32_t pitch = 0;
    uint32_t fmt = 0;
    uint16_t w = 0, h = 0;
    uint32_t bpt = 0; // bytes per texel
    const char* name = "";
};

static inline uint32_t format_bytes(uint32_t f) {
    using nrd::Format;
    switch ((Format)f) {
        case Format::R8_UNORM:
        case Format::R8_UINT: return 1;
        case Format::R16_UINT:
        case Format::R16_UNORM:
        case Format::R16_SFLOAT: return 2;
        case Format::RGBA8_UNORM:
        case Format::RG16_SFLOAT:
        case Format::R32_UINT:
        case Format::R32_SFLOAT:
        case Format::R10_G10_B10_A2_UNORM: return 4;
        case Format::RGBA16_SFLOAT:
        case Format::RGBA16_SNORM:
        case Format::RG32_UINT: return 8;
        case Format::RGBA32_SFLOAT:
        case Format::RGBA32_UINT: return 16;
        default: return 0;
    }
}

// ---- raw texel access -------------------------------------------------------------------------
static inline uint8_t* texel(const Plane& P, int x, int y) { return P.p + (size_t)y * P.pitch + (size_t)x * P.bpt; }
static inline float ld_f32(const Plane& P, int x, int y, int off = 0) { float v; std::memcpy(&v, texel(P, x, y) + off, 4); return v; }
static inline uint32_t ld_u32(const Plane& P, int x, int y, int off = 0) { uint32_t v; std::memcpy(&v, texel(P, x, y) + off, 4); return v; }
static inline uint16_t ld_u16(const Plane& P, int x, int y, int off = 0) { uint16_t v; std::memcpy(&v, texel(P, x, y) + off, 2); return v; }
static inline float ld_h(const Plane& P, int x, int y, int off = 0) { return f16_to_f32(ld_u16(P, x, y, off)); }
static inline f4 ld_h4(const Plane& P, int x, int y, int off = 0) {
    uint16_t v[4];
    std::memcpy(v, texel(P, x, y) + off, 8);
    return {f16_to_f32(v[0]), f16_to_f32(v[1]), f16_to_f32(v[2]), f16_to_f32(v[3])};
}
// luminance of the radiance texel at byte offset `off`: its first fp16 (Y of YCoCg), or Rec.709 of the rgb for the linear-RGB RELAX flavour
static inline float ld_luma(const Plane& P, int x, int y, int off, bool relax) { return (RELAX_LINEAR_RGB && relax) ? luma709(ld_h4(P, x, y, off)) : ld_h(P, x, y, off); }
static inline void st_f32(const Plane& P, int x, int y, float v, int off = 0) { std::memcpy(texel(P, x, y) + off, &v, 4); }
static inline void st_u32(const Plane& P, int x, int y, uint32_t v, int off = 0) { std::memcpy(texel(P, x, y) + off, &v, 4); }
static inline void st_u16(const Plane& P, int x, int y, uint16_t v, int off = 0) { std::memcpy(texel(P, x, y) + off, &v, 2); }
static inline void st_h(const Plane& P, int x, int y, float v, int off = 0) { st_u16(P, x, y, f32_to_f16(clampf(v, -FP16_MAX, FP16_MAX)), off); }
// RGBA16_SNORM texels (the sample's DIRECTIONAL_OCCLUSION data format, Source/NRDSample.cpp:2937): v = max(int16 / 32767, -1)
static inline float sn2f(uint16_t h) { return fmax2((float)(int16_t)h * (1.0f / 32767.0f), -1.0f); }
static inline uint16_t f2sn(float v) { return (uint16_t)(int16_t)floorf(fma_(fmin2(fmax2(v, -1.0f), 1.0f), 32767.0f, 0.5f)); }
static inline f4 ld_sn4(const Plane& P, int x, int y) {
    const uint16_t* t = reinterpret_cast<const uint16_t*>(P.p + (size_t)y * P.pitch + (size_t)x * 8);
    return {sn2f(t[0]), sn2f(t[1]), sn2f(t[2]), sn2f(t[3])};
}
static inline void st_sn4(const Plane& P, int x, int y, f4 v) {
    uint16_t* t = reinterpret_cast<uint16_t*>(P.p + (size_t)y * P.pitch + (size_t)x * 8);
    t[0] = f2sn(v.x);
    t[1] = f2sn(v.y);
    t[2] = f2sn(v.z);
    t[3] = f2sn(v.w);
}
static inline void st_h4(const Plane& P, int x, int y, f4 v, int off = 0) {
    uint16_t h[4] = {f32_to_f16(clampf(v.x, -FP16_MAX, FP16_MAX)), f32_to_f16(clampf(v.y, -FP16_MAX, FP16_MAX)),
                     f32_to_f16(clampf(v.z, -FP16_MAX, FP16_MAX)), f32_to_f16(clampf(v.w, -FP16_MAX, FP16_MAX))};
    std::memcpy(texel(P, x, y) + off, h, 8);
}

// ---- per-frame constants (derived from nrd::CommonSettings; DESIGN.md "frame constants") ---------
struct Consts {
    int W = 0, H = 0;         // rect size of the whole (global) frame
    int Wprev = 0, Hprev = 0; // previous frame's rect
    int resW = 0, resH = 0;   // local plane size
    int yOff = 0;             // global row stored at local row 0 (row tiling), else 0
    int ownY0 = 0, ownY1 = 0; // local rows this instance produces [ownY0, ownY1)
    int prevY0 = 0, prevY1 = 0; // local rows on which the previous frame's planes are current (orc_set_history_rows; default: all stored rows)
    float invW = 0, invH = 0, invWprev = 0, invHprev = 0;
    // orthographic projections (the sample's "Ortho" camera, Source/NRDSample.cpp:1214, :1971): Xv.xy = uv * d + o (no z factor),
    // pj = {m0, m5, m12, m13, 1}; the trailing element of fr / pv / pj is the flag (1 = orthographic) for the helpers below
    float fr[5] = {}, frPrev[5] = {}; // x0, y0, dx, dy : Xv.xy = z * (uv * d + o)
    float pv[5] = {}, pvPrev[5] = {}; // view ray of pixel (px, gy): (pv0 + pv2 * px, pv1 + pv3 * gy, 1)
    float pj[6] = {}, pjPrev[6] = {}; // m0, m5, m8, m9, s (clip.w = s * z)
    bool ortho = false;
    float w2v[9] = {}, w2vPrev[9] = {}, v2w[9] = {}, v2wPrev[9] = {};
    float camDelta[3] = {}; // camera position prev - current (world)
    float unproject = 0, minRectDimMulUnproject = 0;
    float jcx = 0, jcy = 0; // kernel basis -> pixels per pixel of radius: d(px) = +-jcx (T.x - rx T.z), d(py) = +-jcy (T.y - ry T.z) (csrc/nrd_device.h kernel_basis_px)
    float denoisingRange = 0, disocclusionThreshold = 0, splitScreen = 0;
    float disoccAlt = 0;   // CommonSettings::disocclusionThresholdAlternate, blended in per pixel by IN_DISOCCLUSION_THRESHOLD_MIX when ...
    bool mixAvail = false; // ... CommonSettings::isDisocclusionThresholdMixAvailable
    float mvScale[3] = {};
    uint32_t frameIndex = 0;
    uint32_t strandMat = 0xffffffffu; // CommonSettings::strandMaterialID (Source/NRDSample.cpp:3871), 0xffffffff = none
    float strandThickness = 0;        // CommonSettings::strandThickness, world units
    uint32_t camAttachMat = 0xffffffffu; // CommonSettings::cameraAttachedReflectionMaterialID (Source/NRDSample.cpp:3869-3876), 0xffffffff = none
    bool mvWorld = false, confAvail = false, reset = false;
    float rot[64][2] = {};
};

static inline float signed_by(float k, float z) { return u2f(f2u(k) ^ (f2u(z) & 0x80000000u)); } // k * sign(z), exact
// pixel offsets of a view-space tangent pair per pixel of blur radius (csrc/nrd_device.h kernel_basis_px: the depth cancels)
static inline void kernel_basis_px(const Consts& c, float z, float rx, float ry, f3 T, f3 B, float (&j)[4]) {
    if (c.ortho) {
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

bool derive_consts(const nrd::CommonSettings& cs, int resW, int resH, int frameH, int yOff, int ownY0, int ownRows, Consts& c, std::string& err, int histY0 = 0, int histRows = 0);

// view position from uv and (signed) viewZ
static inline f3 reconstruct(const float* fr, float u, float v, float z) {
    float k = fr[4] != 0.0f ? 1.0f : z;
    return {k * (u * fr[2] + fr[0]), k * (v * fr[3] + fr[1]), z};
}
// view position of the centre of pixel (px, gy) at (signed) viewZ
static inline f3 reconstruct_px(const float* pv, float px, float gy, float z) {
    float k = pv[4] != 0.0f ? 1.0f : z;
    return {k * fma_(pv[2], px, pv[0]), k * fma_(pv[3], gy, pv[1]), z};
}
// view position -> uv; false when the point is not in front of the camera
static inline bool project(const float* pj, f3 X, float& u, float& v) {
    if (pj[5] != 0.0f) {
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

// ---- guide texel (8 bytes, round 3; == the guide part of the Blur / PostBlur tap texels) -----------------------------------------------
//   w0 = viewZ rounded to 22 bits | roughness as the 10-bit code of IN_NORMAL_ROUGHNESS. Read back AS ONE FLOAT it is the depth (the
//        roughness code perturbs it by < 2^-13 relative) - every consumer reads it that way
//   w1 = normal x | y << 10 | z << 20, 10 bits per component (n = code * 2/1023 - 1, not re-normalised) | materialID << 30
// written once per pixel by the ClassifyTiles passes from IN_VIEWZ + IN_NORMAL_ROUGHNESS: every bilateral tap decodes its guide with a
// few integer / convert operations instead of an octahedral decode + normalisation, and every pass moves 8 instead of 16 bytes per guide
// access (round 3 measured what these passes pay for: bytes and memory instructions, not arithmetic). The normal arrives as a 10 + 10 bit
// octahedron and the roughness as 10 bits: the texel is as fine as the input; only the depth loses its 10 low mantissa bits.
struct Guide {
    float z; // signed view z (already multiplied by viewZScale), roughness code in its 10 low mantissa bits
    f3 n;
    float roughness;
    uint32_t mat;
    bool sky;
    uint32_t nw; // the texel's normal | material word (normal_cos works on the codes)
};
static const int GUIDE_BYTES = 8;
// Cosine of the angle between two guide normals from their 10-bit CODES: cos = 1 - |n_a - n_b|^2 / 2 (exact for unit vectors). The dot
// product of two quantised, not re-normalised vectors cannot resolve 1 - cos at the 1e-4 level the narrow specular lobes ask for (|n|^2
// is off by up to 2e-3); the squared difference of the codes is exact, zero for equal normals, as fine as the quantisation step.
static inline float normal_dist2(uint32_t nwCentre, uint32_t nw) {
    const float dx = (float)(nw & 1023u) - (float)(nwCentre & 1023u), dy = (float)((nw >> 10) & 1023u) - (float)((nwCentre >> 10) & 1023u),
                dz = (float)((nw >> 20) & 1023u) - (float)((nwCentre >> 20) & 1023u);
    return fma_(dz, dz, fma_(dy, dy, dx * dx));
}
static inline float normal_cos(uint32_t nwCentre, uint32_t nw) { return fma_(normal_dist2(nwCentre, nw), -NORMAL_D2_TO_1MCOS, 1.0f); }
static inline uint32_t guide_qn10(float v) { return (uint32_t)floorf(clampf(fma_(v, 511.5f, 512.0f), 0.0f, 1023.0f)); }
// A pixel without geometry (|viewZ| beyond the range, Inf, NaN - e.g. a 0xFF-filled "no hit" plane, whose bits would wrap in the rounding)
// stores one canonical finite depth; the return value says whether the STORED depth has geometry - the test every consumer applies
// (csrc/nrd_device.h encode_guide)
static const uint32_t GUIDE_SKY_DEPTH = 0x7F7FFC00u;
static inline bool store_guide(const Plane& G, int x, int y, float z, uint32_t packedNR, float range) {
    NormalRoughness nr = unpack_normal_roughness(packedNR);
    const uint32_t code = (packedNR >> 20) & 1023u;
    uint32_t w0 = ((f2u(z) + 0x200u) & 0xFFFFFC00u) | code;
    const bool geo = absf(z) <= range && absf(u2f(w0)) <= range;
    if (!geo)
        w0 = GUIDE_SKY_DEPTH | code;
    uint32_t w1 = guide_qn10(nr.n.x) | (guide_qn10(nr.n.y) << 10) | (guide_qn10(nr.n.z) << 20) | (nr.materialID << 30);
    st_u32(G, x, y, w0, 0);
    st_u32(G, x, y, w1, 4);
    return geo;
}
static inline Guide decode_guide_words(uint32_t w0, uint32_t w1, float range) {
    Guide g;
    g.nw = w1;
    g.z = u2f(w0);
    g.roughness = (float)(w0 & 1023u) * (1.0f / 1023.0f);
    const float s = 2.0f / 1023.0f;
    g.n = {fma_((float)(w1 & 1023u), s, -1.0f), fma_((float)((w1 >> 10) & 1023u), s, -1.0f), fma_((float)((w1 >> 20) & 1023u), s, -1.0f)};
    g.mat = w1 >> 30;
    g.sky = !(absf(g.z) <= range);
    return g;
}
static inline Guide load_guide(const Plane& G, int x, int y, float range) { return decode_guide_words(ld_u32(G, x, y, 0), ld_u32(G, x, y, 4), range); }
static inline float guide_roughness(const Plane& G, int x, int y) { return (float)(ld_u32(G, x, y, 0) & 1023u) * (1.0f / 1023.0f); }
// material comparison: ids differ and the larger one takes part in material-aware filtering
static inline bool material_mismatch(uint32_t a, uint32_t b, uint32_t minMaterial) { return a != b && (a > b ? a : b) >= minMaterial; }

// hard bound (pixels, rows and columns) on how far a bilateral tap may land from its centre; also the halo a row-tiled
// instance needs for the pass
struct ReblurReach {
    int pre, blur, post;
};
static inline ReblurReach reblur_reach(const nrd::ReblurSettings& s) {
    ReblurReach r;
    r.pre = (int)(fmax2(s.diffusePrepassBlurRadius, s.specularPrepassBlurRadius) * 1.1f) + 3;
    r.blur = (int)((s.maxBlurRadius + s.minBlurRadius) * 1.1f) + 3;
    r.post = (int)((s.maxBlurRadius + s.minBlurRadius) * 2.2f) + 3;
    return r;
}

// ---- denoiser bookkeeping -----------------------------------------------------------------------
enum class Kind { REBLUR, RELAX, SIGMA, REFERENCE };

struct PoolPlane {
    const char* name;
    uint32_t fmt;
    uint32_t bpt;
    uint16_t downsample;
};

struct Instance;
struct DenoiserState;

struct Pass {
    const char* name;
    const char* kernel;
    uint16_t haloRows;
    float bytesPerPixel;
    std::vector<uint32_t> written; // (pool << 16) | index ; pool 2 = slot
    std::vector<uint32_t> read;
    std::function<void(Instance&, DenoiserState&, const Consts&, int y0, int y1)> run; // rows [y0,y1) local
    bool tileGrid = false; // pass iterates over 16x16 tiles rather than pixels
    // row tiling (nrdhip_dispatch_info::read_rows / flags, csrc/nrdhip.cpp Dispatch): planes of `read` fetched at the pixel's own position
    // only / previous-frame state fetched at motion-displaced positions / planes reached less far into than haloRows; a pointwise pass
    // that runs on every stored row of a band (the ClassifyTiles passes)
    std::vector<uint32_t> own, reprojected;
    std::vector<std::pair<uint32_t, uint16_t>> reach;
    bool allRows = false;
    std::vector<std::pair<uint32_t, uint32_t>> prefix; // nrdhip_dispatch_info::written_prefix: {written tap-texel plane, the guide plane its texels start with}
};

struct DenoiserState {
    uint32_t identifier = 0;
    nrd::Denoiser denoiser = nrd::Denoiser::MAX_NUM;
    Kind kind = Kind::REFERENCE;
    bool hasDiff = false, hasSpec = false, sh = false, occlusion = false, translucency = false;
    bool dirOcc = false; // REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION: one {direction * h, h} texel in / out, filtered as SH0 = {h,0,0,h} + SH1 = direction * h
    int nsig = 0;
    uint32_t permBase = 0, transBase = 0; // first plane index in the instance pools
    uint32_t frameCounter = 0;            // denoise calls so far (ping-pong selector)
    bool historyValid = false;
    uint32_t framesSinceReset = 0; // frames accumulated since the last history reset
    nrd::ReblurSettings reblur;
    nrd::RelaxSettings relax;
    nrd::SigmaSettings sigma;
    nrd::ReferenceSettings reference;
    std::vector<Pass> passes; // rebuilt every denoise call
};

struct Instance {
    int resW = 0, resH = 0, frameH = 0, yOff = 0, ownY0 = 0, ownRows = 0;
    int histY0 = 0, histRows = 0; // orc_set_history_rows (0 rows = all)
    uint32_t flags = 0;
    int threads = 1;
    nrd::CommonSettings common;
    bool commonSet = false;
    std::vector<DenoiserState> denoisers;
    std::vector<PoolPlane> permDesc, transDesc;
    std::vector<Plane> perm, trans;
    std::vector<std::vector<uint8_t>> owned; // internal storage when pools are not external
    Plane slots[(size_t)nrd::ResourceType::MAX_NUM];
    std::string error;
};

// pass-graph builders (one per denoiser family)
void reference_describe(DenoiserState& d, std::vector<PoolPlane>& perm, std::vector<PoolPlane>& trans);
void reference_build(Instance& I, DenoiserState& d);
void reblur_describe(DenoiserState& d, std::vector<PoolPlane>& perm, std::vector<PoolPlane>& trans);
void reblur_build(Instance& I, DenoiserState& d);
void sigma_describe(DenoiserState& d, std::vector<PoolPlane>& perm, std::vector<PoolPlane>& trans);
void sigma_build(Instance& I, DenoiserState& d);
void relax_describe(DenoiserState& d, std::vector<PoolPlane>& perm, std::vector<PoolPlane>& trans);
void relax_build(Instance& I, DenoiserState& d);

static inline uint32_t enc_perm(uint32_t i) { return (0u << 16) | i; }
static inline uint32_t enc_trans(uint32_t i) { return (1u << 16) | i; }
static inline uint32_t enc_slot(nrd::ResourceType t) { return (2u << 16) | (uint32_t)t; }

// 8-tap Poisson disk (unit radius) + per-tap Gaussian-like weight; frozen table
extern const float g_poisson8[8][3];

} // namespace orc
