// orc_sigma.cpp - CPU ORACLE of SIGMA_SHADOW / SIGMA_SHADOW_TRANSLUCENCY (test infrastructure; PARITY UNPINNED
// vs upstream NRD whose sources are absent from the reference tree).
//
// Contract followed (reference call sites, relative to /root/reference):
//   slots & formats .... Source/NRDSample.cpp:479-481 (bind), :2945, :2978, :2981, :2984 (R16F penumbra, RGBA8 translucency/shadow)
//   settings ........... Source/NRDSample.cpp:4072-4082 (lightDirection), :2175-2177 (maxStabilizedFrameNum)
//   input encoding ..... Shaders/TraceOpaque.cs.hlsl:779-804: penumbra = distance to occluder * tan(light angular radius),
//                        "no occluder" keeps the INF-summed distance (packed as FP16 max); translucency .x = lit flag, .yzw = colour
//   output decoding .... Shaders/Composition.cs.hlsl:60-64: SIGMA_BackEnd_UnpackShadow(x) = x^2, .x shadow, .yzw coloured shadow
// Pass graph: ClassifyTiles(+guide) -> SmoothTiles -> Blur -> PostBlur -> TemporalStabilization(+split screen).
#include "orc_core.h"

namespace orc {

namespace {

enum Perm { P_GUIDE_A, P_GUIDE_B, P_HIST_A, P_HIST_B };
enum Trans { T_TILES, T_TILES_SMOOTH, T_SHADOW1, T_PEN1, T_SHADOW2 };

const float MAX_PIXEL_RADIUS = 48.0f;
const int BLUR_REACH = 56; // (int)(48 * 1.1) + 3
const float PREV_NORMAL_COS = 0.7f;
const float STAB_SIGMA_SCALE = 2.0f;

struct Ctx {
    Instance& I;
    DenoiserState& d;
    const Consts& c;
    int cur;
    const Plane& perm(int i) const { return I.perm[d.permBase + i]; }
    const Plane& trans(int i) const { return I.trans[d.transBase + i]; }
    const Plane& slot(nrd::ResourceType t) const { return I.slots[(size_t)t]; }
};

static inline f4 mul4(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline f4 fma4(f4 a, float s, f4 c) { return {fma_(a.x, s, c.x), fma_(a.y, s, c.y), fma_(a.z, s, c.z), fma_(a.w, s, c.w)}; }

// visibility signal of a raw input texel: lit -> 1, shadowed -> (0, translucency)
static inline f4 input_visibility(const Ctx& k, int x, int y, float pen) {
    if (pen >= FP16_MAX)
        return {1, 1, 1, 1};
    if (!k.d.translucency)
        return {0, 0, 0, 0};
    const uint8_t* t = texel(k.slot(nrd::ResourceType::IN_TRANSLUCENCY), x, y);
    return {0.0f, (float)t[1] * (1.0f / 255.0f), (float)t[2] * (1.0f / 255.0f), (float)t[3] * (1.0f / 255.0f)};
}

static inline uint32_t encode_shadow(f4 v) {
    uint32_t r = 0;
    float c[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < 4; i++)
        r |= (uint32_t)floorf(fma_(sqrtf(sat(c[i])), 255.0f, 0.5f)) << (8 * i);
    return r;
}
static inline f4 decode_shadow(uint32_t p) {
    float c[4];
    for (int i = 0; i < 4; i++) {
        float b = (float)((p >> (8 * i)) & 255u) * (1.0f / 255.0f);
        c[i] = b * b;
    }
    return {c[0], c[1], c[2], c[3]};
}

void classify_tiles(Instance& I, DenoiserState& d, const Consts& c, int ty0, int ty1) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const Plane& inZ = k.slot(nrd::ResourceType::IN_VIEWZ);
    const Plane& inNR = k.slot(nrd::ResourceType::IN_NORMAL_ROUGHNESS);
    const Plane& inPen = k.slot(nrd::ResourceType::IN_PENUMBRA);
    const Plane& G = k.perm(P_GUIDE_A + k.cur);
    const Plane& T = k.trans(T_TILES);
    float zs = I.common.viewZScale;
    int tilesX = (c.W + 15) / 16;
    for (int ty = ty0; ty < ty1; ty++)
        for (int tx = 0; tx < tilesX; tx++) {
            uint32_t flags = 0;
            float maxR = 0.0f;
            for (int j = 0; j < 16; j++)
                for (int i = 0; i < 16; i++) {
                    int x = tx * 16 + i, y = ty * 16 + j;
                    if (x >= c.W || y >= c.resH || y + c.yOff >= c.H || y + c.yOff < 0)
                        continue;
                    float z = ld_f32(inZ, x, y) * zs;
                    if (!store_guide(G, x, y, z, ld_u32(inNR, x, y), c.denoisingRange))
                        continue;
                    float pen = ld_h(inPen, x, y);
                    if (pen >= FP16_MAX)
                        flags |= 2u;
                    else {
                        flags |= 1u;
                        maxR = fmax2(maxR, fmin2(pen * rcp_(c.unproject * (c.ortho ? 1.0f : absf(z))), 255.0f));
                    }
                }
            uint32_t r = (uint32_t)floorf(maxR + 0.999f);
            r = r > 255u ? 255u : r;
            st_u16(T, tx, ty, (uint16_t)(flags | (r << 8)));
        }
}

void smooth_tiles(Instance& I, DenoiserState& d, const Consts& c, int ty0, int ty1) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const Plane& T = k.trans(T_TILES);
    const Plane& S = k.trans(T_TILES_SMOOTH);
    int tilesX = (c.W + 15) / 16;
    int tilesY = (c.resH + 15) / 16;
    for (int ty = ty0; ty < ty1; ty++)
        for (int tx = 0; tx < tilesX; tx++) {
            uint32_t flags = 0, r = 0;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    int x = tx + i, y = ty + j;
                    if (x < 0 || x >= tilesX || y < 0 || y >= tilesY)
                        continue;
                    uint32_t t = ld_u16(T, x, y);
                    flags |= t & 3u;
                    r = std::max(r, t >> 8);
                }
            st_u16(S, tx, ty, (uint16_t)((flags == 3u ? 1u : 0u) | (r << 8)));
        }
}

// shared by Blur (pass 0: raw inputs) and PostBlur (pass 1: Shadow1/Pen1)
void blur(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1, int pass) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const nrd::SigmaSettings& s = d.sigma;
    const Plane& G = k.perm(P_GUIDE_A + k.cur);
    const Plane& S = k.trans(T_TILES_SMOOTH);
    const Plane& inPen = pass == 0 ? k.slot(nrd::ResourceType::IN_PENUMBRA) : k.trans(T_PEN1);
    const Plane& inSh = k.trans(T_SHADOW1);
    const Plane& outSh = pass == 0 ? k.trans(T_SHADOW1) : k.trans(T_SHADOW2);
    const Plane& outPen = k.trans(T_PEN1);
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c.W; x++) {
            float z = ld_f32(G, x, y, 0);
            if (!(absf(z) <= c.denoisingRange)) {
                st_h4(outSh, x, y, {0, 0, 0, 0});
                if (pass == 0)
                    st_h(outPen, x, y, 0.0f);
                continue;
            }
            float absZ = absf(z);
            float pen = ld_h(inPen, x, y);
            bool lit = pass == 0 ? pen >= FP16_MAX : !(pen > 0.0f);
            f4 center = pass == 0 ? input_visibility(k, x, y, pen) : ld_h4(inSh, x, y);
            uint32_t tile = ld_u16(S, x / 16, y / 16);
            if (!(tile & 1u)) {
                st_h4(outSh, x, y, center);
                if (pass == 0)
                    st_h(outPen, x, y, lit ? 0.0f : pen);
                continue;
            }
            float pixelWorld = c.unproject * (c.ortho ? 1.0f : absZ);
            float radiusPx = lit ? (float)(tile >> 8) : pen * rcp_(pixelWorld);
            radiusPx = fmin2(radiusPx, MAX_PIXEL_RADIUS);
            int gy0 = y + c.yOff;
            Guide g = load_guide(G, x, y, c.denoisingRange);
            const float rx = fma_(c.pv[2], (float)x, c.pv[0]), ry = fma_(c.pv[3], (float)gy0, c.pv[1]); // view ray (orthographic: view-space xy)
            f3 Xv = {(c.ortho ? 1.0f : z) * rx, (c.ortho ? 1.0f : z) * ry, z};
            f3 Nv = rot3(c.w2v, g.n);
            float frustumSize = c.minRectDimMulUnproject * (c.ortho ? 1.0f : absZ);
            float geoA = rcp_(s.planeDistanceSensitivity * frustumSize);
            float gax = Nv.x * c.pv[2] * geoA, gay = Nv.y * c.pv[3] * geoA;
            // plane distance of a tap = |zs * (ga0 + gax px + gay gy) + geoB|; orthographic: |zs * geoB + (ga0 + gax px + gay gy)|
            float ga0 = c.ortho ? (fma_(Nv.x, c.pv[0], Nv.y * c.pv[1]) - dot3(Nv, Xv)) * geoA : fma_(Nv.x, c.pv[0], fma_(Nv.y, c.pv[1], Nv.z)) * geoA;
            float geoB = c.ortho ? Nv.z * geoA : -dot3(Nv, Xv) * geoA;
            f3 T, B;
            basis3(Nv, T, B);
            float ju[4]; // the kernel basis in pixels per pixel of radius (orc_core.h kernel_basis_px)
            kernel_basis_px(c, z, rx, ry, T, B, ju);
            float jtx = ju[0] * radiusPx, jty = ju[1] * radiusPx;
            float jbx = ju[2] * radiusPx, jby = ju[3] * radiusPx;
            bool perPixel = pass == 0; // Blur rotates per pixel, PostBlur per frame
            uint32_t h = hash_px(perPixel ? (uint32_t)x : 0u, perPixel ? (uint32_t)gy0 : 0u, c.frameIndex, 17u + (uint32_t)pass);
            float rc = c.rot[h & 63u][0], rs = c.rot[h & 63u][1];
            { // rotation folded into the Jacobian (J . R): the taps then are the unrotated disk
                const float a = fma_(rc, jtx, rs * jbx), b = fma_(rc, jbx, -(rs * jtx));
                const float cc = fma_(rc, jty, rs * jby), d = fma_(rc, jby, -(rs * jty));
                jtx = a;
                jbx = b;
                jty = cc;
                jby = d;
            }
            f4 sum = center;
            float wsum = 1.0f;
            float penSum = lit ? 0.0f : pen, penW = lit ? 0.0f : 1.0f;
            if (radiusPx > 0.0f)
                for (int t = 0; t < 8; t++) {
                    const float ox = g_poisson8[t][0], oy = g_poisson8[t][1]; // rotation folded into the Jacobian above
                    float fpx = floorf(fma_(ox, jtx, fma_(oy, jbx, (float)x + 0.5f)));
                    float fpy = floorf(fma_(ox, jty, fma_(oy, jby, (float)gy0 + 0.5f)));
                    if (!(fpx >= 0.0f && fpx < (float)c.W && fpy >= 0.0f && fpy < (float)c.H))
                        continue;
                    int px = (int)fpx, gy = (int)fpy, py = gy - c.yOff;
                    int ddx = px - x, ddy = gy - gy0;
                    if (ddx > BLUR_REACH || -ddx > BLUR_REACH || ddy > BLUR_REACH || -ddy > BLUR_REACH)
                        continue;
                    if (py < 0 || py >= c.resH)
                        continue;
                    float zs = ld_f32(G, px, py, 0);
                    if (!(absf(zs) <= c.denoisingRange))
                        continue;
                    float ga = fma_(gax, fpx, fma_(gay, fpy, ga0));
                    float w = g_poisson8[t][2] * smoothstep01(1.0f - absf(c.ortho ? fma_(zs, geoB, ga) : fma_(zs, ga, geoB)));
                    float ps = ld_h(inPen, px, py);
                    bool lits = pass == 0 ? ps >= FP16_MAX : !(ps > 0.0f);
                    f4 sv = pass == 0 ? input_visibility(k, px, py, ps) : ld_h4(inSh, px, py);
                    sum = fma4(sv, w, sum);
                    wsum += w;
                    if (!lits) {
                        penSum = fma_(ps, w, penSum);
                        penW += w;
                    }
                }
            st_h4(outSh, x, y, mul4(sum, rcp_(wsum)));
            if (pass == 0)
                st_h(outPen, x, y, penW > 0.0f ? penSum * rcp_(penW) : 0.0f);
        }
}

void temporal_stabilization(Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
    Ctx k{I, d, c, (int)(d.frameCounter & 1)};
    const nrd::SigmaSettings& s = d.sigma;
    const Plane& G = k.perm(P_GUIDE_A + k.cur);
    const Plane& GP = k.perm(P_GUIDE_A + (k.cur ^ 1));
    const Plane& HP = k.perm(P_HIST_A + (k.cur ^ 1));
    const Plane& HC = k.perm(P_HIST_A + k.cur);
    const Plane& SH = k.trans(T_SHADOW2);
    const Plane& ST = k.trans(T_TILES_SMOOTH);
    const Plane& MV = k.slot(nrd::ResourceType::IN_MV);
    const Plane& OUT = k.slot(nrd::ResourceType::OUT_SHADOW_TRANSLUCENCY);
    const Plane& inPen = k.slot(nrd::ResourceType::IN_PENUMBRA);
    bool historyOk = d.historyValid && !c.reset;
    float maxStab = (float)std::min<uint32_t>(s.maxStabilizedFrameNum, nrd::SIGMA_MAX_HISTORY_FRAME_NUM);
    auto store_out = [&](int x, int y, uint32_t packed) {
        if (OUT.bpt == 1)
            *texel(OUT, x, y) = (uint8_t)(packed & 255u);
        else
            st_u32(OUT, x, y, packed);
    };
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c.W; x++) {
            int gy0 = y + c.yOff;
            float u = ((float)x + 0.5f) * c.invW, v = ((float)gy0 + 0.5f) * c.invH;
            bool split = u < c.splitScreen;
            float z = ld_f32(G, x, y, 0);
            if (!(absf(z) <= c.denoisingRange)) {
                st_u32(HC, x, y, 0);
                store_out(x, y, split ? encode_shadow(input_visibility(k, x, y, ld_h(inPen, x, y))) : 0u);
                continue;
            }
            f4 cur = ld_h4(SH, x, y);
            // tile check: no penumbra in this tile or its 8 neighbours (SmoothTiles) -> every texel of the 5x5 window is an
            // unfiltered lit / umbra value: nothing to stabilize, the history simply follows the signal
            if (!(ld_u16(ST, x / 16, y / 16) & 1u)) {
                uint32_t packed = encode_shadow(cur);
                st_u32(HC, x, y, packed);
                store_out(x, y, split ? encode_shadow(input_visibility(k, x, y, ld_h(inPen, x, y))) : packed);
                continue;
            }
            // 5x5 moments per channel
            float m1[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0};
            for (int j = -2; j <= 2; j++)
                for (int i = -2; i <= 2; i++) {
                    int px = x + i, py = y + j, gy = py + c.yOff;
                    f4 f = cur;
                    if (px >= 0 && px < c.W && gy >= 0 && gy < c.H && py >= 0 && py < c.resH) {
                        float zt = ld_f32(G, px, py, 0);
                        if (absf(zt) <= c.denoisingRange)
                            f = ld_h4(SH, px, py);
                    }
                    float fc[4] = {f.x, f.y, f.z, f.w};
                    for (int ch = 0; ch < 4; ch++) {
                        m1[ch] += fc[ch];
                        m2[ch] = fma_(fc[ch], fc[ch], m2[ch]);
                    }
                }
            // surface-motion reprojection with plane-distance occlusion test
            Guide g = load_guide(G, x, y, c.denoisingRange);
            f3 Xv = reconstruct_px(c.pv, (float)x, (float)gy0, z);
            f4 mvRaw = ld_h4(MV, x, y);
            f3 Xw = rot3(c.v2w, Xv);
            f3 cd = {c.camDelta[0], c.camDelta[1], c.camDelta[2]};
            float su, sv;
            f3 XvPrev;
            bool uvOk = true;
            if (c.mvWorld) {
                f3 XwPrev = add3(Xw, {mvRaw.x * c.mvScale[0], mvRaw.y * c.mvScale[1], mvRaw.z * c.mvScale[2]});
                XvPrev = rot3(c.w2vPrev, sub3(XwPrev, cd));
                uvOk = project(c.pjPrev, XvPrev, su, sv);
            } else {
                su = fma_(mvRaw.x, c.mvScale[0], u);
                sv = fma_(mvRaw.y, c.mvScale[1], v);
                if (c.mvScale[2] != 0.0f)
                    XvPrev = reconstruct(c.frPrev, su, sv, fma_(mvRaw.z, c.mvScale[2], z));
                else
                    XvPrev = rot3(c.w2vPrev, sub3(Xw, cd));
            }
            f4 hist = cur;
            bool have = false;
            if (historyOk && uvOk) {
                f3 NvPrev = rot3(c.w2vPrev, g.n);
                float thrBase = c.disocclusionThreshold;
                if (c.mixAvail) // per-pixel blend toward disocclusionThresholdAlternate (IN_DISOCCLUSION_THRESHOLD_MIX, R8_UNORM)
                    thrBase = lerpf(thrBase, c.disoccAlt, (float)*texel(k.slot(nrd::ResourceType::IN_DISOCCLUSION_THRESHOLD_MIX), x, y) * (1.0f / 255.0f));
                float threshold = thrBase * c.minRectDimMulUnproject * (c.ortho ? 1.0f : absf(XvPrev.z));
                float px = fma_(su, (float)c.Wprev, -0.5f), py = fma_(sv, (float)c.Hprev, -0.5f);
                float fx0 = floorf(px), fy0 = floorf(py);
                float fx = px - fx0, fy = py - fy0;
                bool sane = fx0 >= -2.0f && fx0 <= (float)c.Wprev + 1.0f && fy0 >= -2.0f && fy0 <= (float)c.Hprev + 1.0f;
                if (sane) {
                    int ix = (int)fx0, iy = (int)fy0;
                    float bw[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
                    float planeRef = dot3(NvPrev, XvPrev);
                    float g0 = c.ortho ? fma_(NvPrev.x, c.pvPrev[0], NvPrev.y * c.pvPrev[1]) : fma_(NvPrev.x, c.pvPrev[0], fma_(NvPrev.y, c.pvPrev[1], NvPrev.z));
                    float gx = NvPrev.x * c.pvPrev[2], gyc = NvPrev.y * c.pvPrev[3];
                    f4 sum = {0, 0, 0, 0};
                    float wsum = 0.0f;
                    for (int i = 0; i < 4; i++) {
                        int tx = ix + (i & 1), gy = iy + (i >> 1), ty = gy - c.yOff;
                        if (tx < 0 || tx >= c.Wprev || gy < 0 || gy >= c.Hprev || ty < c.prevY0 || ty >= c.prevY1)
                            continue;
                        Guide gp = load_guide(GP, tx, ty, c.denoisingRange);
                        if (gp.sky)
                            continue;
                        float lin = fma_(gx, (float)tx, fma_(gyc, (float)gy, g0));
                        float plane = c.ortho ? fma_(gp.z, NvPrev.z, lin) : gp.z * lin;
                        if (!(absf(plane - planeRef) <= threshold) || !(dot3(g.n, gp.n) > PREV_NORMAL_COS))
                            continue;
                        sum = fma4(decode_shadow(ld_u32(HP, tx, ty)), bw[i], sum);
                        wsum += bw[i];
                    }
                    if (wsum > 0.0f) {
                        hist = mul4(sum, rcp_(wsum));
                        have = true;
                    }
                }
            }
            float w = have ? maxStab / (1.0f + maxStab) : 0.0f;
            float hc[4] = {hist.x, hist.y, hist.z, hist.w}, cc[4] = {cur.x, cur.y, cur.z, cur.w}, o[4];
            for (int ch = 0; ch < 4; ch++) {
                float a = m1[ch] * (1.0f / 25.0f), b = m2[ch] * (1.0f / 25.0f);
                float sigma = sqrtf(fmax2(fma_(-a, a, b), 0.0f)) * STAB_SIGMA_SCALE;
                float hcl = clampf(hc[ch], a - sigma, a + sigma);
                o[ch] = lerpf(cc[ch], hcl, w);
            }
            uint32_t packed = encode_shadow({o[0], o[1], o[2], o[3]});
            st_u32(HC, x, y, packed);
            store_out(x, y, split ? encode_shadow(input_visibility(k, x, y, ld_h(inPen, x, y))) : packed);
        }
}

} // namespace

void sigma_describe(DenoiserState&, std::vector<PoolPlane>& perm, std::vector<PoolPlane>& trans) {
    perm.push_back({"SIGMA::Guide_A", (uint32_t)nrd::Format::RG32_UINT, 8, 1});
    perm.push_back({"SIGMA::Guide_B", (uint32_t)nrd::Format::RG32_UINT, 8, 1});
    perm.push_back({"SIGMA::History_A", (uint32_t)nrd::Format::RGBA8_UNORM, 4, 1});
    perm.push_back({"SIGMA::History_B", (uint32_t)nrd::Format::RGBA8_UNORM, 4, 1});
    trans.push_back({"SIGMA::Tiles", (uint32_t)nrd::Format::R16_UINT, 2, 16});
    trans.push_back({"SIGMA::SmoothTiles", (uint32_t)nrd::Format::R16_UINT, 2, 16});
    trans.push_back({"SIGMA::Shadow1", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, 1});
    trans.push_back({"SIGMA::Penumbra1", (uint32_t)nrd::Format::R16_SFLOAT, 2, 1});
    trans.push_back({"SIGMA::Shadow2", (uint32_t)nrd::Format::RGBA16_SFLOAT, 8, 1});
}

void sigma_build(Instance& I, DenoiserState& d) {
    using RT = nrd::ResourceType;
    int cur = (int)(d.frameCounter & 1);
    uint32_t pb = d.permBase, tb = d.transBase;
    auto P = [&](int i) { return enc_perm(pb + i); };
    auto T = [&](int i) { return enc_trans(tb + i); };
    float tr = d.translucency ? 4.0f : 0.0f;
    const float GB = (float)GUIDE_BYTES;
    {
        Pass p;
        p.name = "SIGMA::ClassifyTiles";
        p.kernel = "nrd_sigma_classify_tiles";
        p.haloRows = 0;
        p.bytesPerPixel = 4 + 4 + 2 + GB + 2.0f / 256.0f;
        p.read = {enc_slot(RT::IN_VIEWZ), enc_slot(RT::IN_NORMAL_ROUGHNESS), enc_slot(RT::IN_PENUMBRA)};
        p.written = {P(P_GUIDE_A + cur), T(T_TILES)};
        p.tileGrid = true;
        p.allRows = true;
        p.run = classify_tiles;
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "SIGMA::SmoothTiles";
        p.kernel = "nrd_sigma_smooth_tiles";
        p.haloRows = 16;
        p.bytesPerPixel = 4.0f / 256.0f;
        p.read = {T(T_TILES)};
        p.written = {T(T_TILES_SMOOTH)};
        p.tileGrid = true;
        p.run = smooth_tiles;
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "SIGMA::Blur";
        p.kernel = "nrd_sigma_blur";
        p.haloRows = (uint16_t)BLUR_REACH;
        p.bytesPerPixel = GB + 2 + tr + 8 + 2;
        p.read = {P(P_GUIDE_A + cur), T(T_TILES_SMOOTH), enc_slot(RT::IN_PENUMBRA)};
        if (d.translucency)
            p.read.push_back(enc_slot(RT::IN_TRANSLUCENCY));
        p.written = {T(T_SHADOW1), T(T_PEN1)};
        p.run = [](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) { blur(I, d, c, y0, y1, 0); };
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "SIGMA::PostBlur";
        p.kernel = "nrd_sigma_post_blur";
        p.haloRows = (uint16_t)BLUR_REACH;
        p.bytesPerPixel = GB + 8 + 2 + 8;
        p.read = {P(P_GUIDE_A + cur), T(T_TILES_SMOOTH), T(T_SHADOW1), T(T_PEN1)};
        p.written = {T(T_SHADOW2)};
        p.run = [](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) { blur(I, d, c, y0, y1, 1); };
        d.passes.push_back(p);
    }
    {
        Pass p;
        p.name = "SIGMA::TemporalStabilization";
        p.kernel = "nrd_sigma_temporal_stabilization";
        p.haloRows = 2;
        p.bytesPerPixel = GB + GB + 8 + 8 + 4 + 4 + 4;
        p.read = {P(P_GUIDE_A + cur), P(P_GUIDE_A + (cur ^ 1)), P(P_HIST_A + (cur ^ 1)), T(T_SHADOW2), T(T_TILES_SMOOTH), enc_slot(RT::IN_MV), enc_slot(RT::IN_PENUMBRA)};
        if (I.common.isDisocclusionThresholdMixAvailable)
            p.read.push_back(enc_slot(RT::IN_DISOCCLUSION_THRESHOLD_MIX));
        if (d.translucency)
            p.read.push_back(enc_slot(RT::IN_TRANSLUCENCY));
        p.written = {P(P_HIST_A + cur), enc_slot(RT::OUT_SHADOW_TRANSLUCENCY)};
        p.reprojected = {P(P_GUIDE_A + (cur ^ 1)), P(P_HIST_A + (cur ^ 1))};
        p.run = temporal_stabilization;
        d.passes.push_back(p);
    }
}

} // namespace orc
