// nrd_sigma.hip - SIGMA_SHADOW / SIGMA_SHADOW_TRANSLUCENCY passes as gfx950 HIP kernels, plus the REFERENCE accumulator.
//
// Replaces the SIGMA / REFERENCE HLSL passes of the reference's absent External/NRD submodule behind
// nrd::Integration::Denoise (Source/NRDSample.cpp:4082 shadow denoising, :4224 reference accumulation).
// Inputs: IN_PENUMBRA R16F / IN_TRANSLUCENCY RGBA8 (Shaders/TraceOpaque.cs.hlsl:800-804); output RGBA8, sqrt-encoded
// (Shaders/Composition.cs.hlsl:60-64 squares it). Pass graph: ClassifyTiles(+guide) -> SmoothTiles -> Blur -> PostBlur ->
// TemporalStabilization. Same 16x16 XCD-swizzled tiling and 16-byte pre-decoded guide texel as REBLUR; the 5x5 clamp stencil
// of the stabilization pass stages its 20x20 RGBA tile in LDS.
#include "nrd_kernels.h"

namespace nrdhip {

NRD_KERNELS_BEGIN

constexpr float MAX_PIXEL_RADIUS = 48.0f;
constexpr int BLUR_REACH = 56; // (int)(48 * 1.1) + 3
constexpr float PREV_NORMAL_COS = 0.7f;
constexpr float STAB_SIGMA_SCALE = 2.0f;

// visibility of an input texel from its penumbra and (SIGMA_SHADOW_TRANSLUCENCY) its IN_TRANSLUCENCY texel `t`
NRD_DEV f4 visibility_of(const SigmaParams& p, float pen, uint32_t t) {
    if (pen >= NRD_FP16_MAX)
        return {1, 1, 1, 1};
    if (!p.translucency)
        return {0, 0, 0, 0};
    return {0.0f, (float)((t >> 8) & 255u) * (1.0f / 255.0f), (float)((t >> 16) & 255u) * (1.0f / 255.0f), (float)(t >> 24) * (1.0f / 255.0f)};
}
NRD_DEV f4 input_visibility(const SigmaParams& p, int x, int y, float pen) {
    if (pen >= NRD_FP16_MAX || !p.translucency)
        return visibility_of(p, pen, 0u);
    return visibility_of(p, pen, ld<uint32_t>(p.inTransl, x, y, 4));
}

NRD_DEV uint32_t encode_shadow(f4 v) {
    uint32_t r = (uint32_t)__builtin_floorf(fma_(__builtin_sqrtf(sat(v.x)), 255.0f, 0.5f));
    r |= (uint32_t)__builtin_floorf(fma_(__builtin_sqrtf(sat(v.y)), 255.0f, 0.5f)) << 8;
    r |= (uint32_t)__builtin_floorf(fma_(__builtin_sqrtf(sat(v.z)), 255.0f, 0.5f)) << 16;
    r |= (uint32_t)__builtin_floorf(fma_(__builtin_sqrtf(sat(v.w)), 255.0f, 0.5f)) << 24;
    return r;
}
NRD_DEV f4 decode_shadow(uint32_t p) {
    float a = (float)(p & 255u) * (1.0f / 255.0f), b = (float)((p >> 8) & 255u) * (1.0f / 255.0f);
    float c = (float)((p >> 16) & 255u) * (1.0f / 255.0f), d = (float)(p >> 24) * (1.0f / 255.0f);
    return {a * a, b * b, c * c, d * d};
}

__global__ __launch_bounds__(256) void k_sigma_classify_tiles(const SigmaParams p) {
    __shared__ float smax[4];
    const FrameConsts& c = p.c;
    // a streaming pass without neighbour reads: plain 2-D grid of tiles (like REBLUR's ClassifyTiles: the XCD traversal's ~700 cycles of
    // scalar index arithmetic buy it nothing), and the penumbra texel travels with depth and normal instead of after the range test
    const int tx = (int)blockIdx.x, ty = (int)blockIdx.y + c.tileY0;
    int x = tx * 16 + (int)threadIdx.x, y = ty * 16 + (int)threadIdx.y;
    bool valid = x < c.W && y < c.resH && (y + c.yOff) < c.H && (y + c.yOff) >= 0;
    int shadowed = 0, lit = 0;
    float r = 0.0f;
    if (valid) {
        const float zRaw = ld<float>(p.inZ, x, y, 4);
        const uint32_t nr = ld<uint32_t>(p.inNR, x, y, 4);
        const uint16_t penRaw = ld<uint16_t>(p.inPen, x, y, 2);
        float z = zRaw * c.viewZScale;
        bool geo;
        st<uint2>(p.guide, x, y, GUIDE_BYTES, encode_guide(z, nr, c.denoisingRange, geo));
        if (geo) {
            float pen = h2f(penRaw);
            if (pen >= NRD_FP16_MAX)
                lit = 1;
            else {
                shadowed = 1;
                r = fmin2(pen * rcp_(c.unproject * zpersp(absf(z))), 255.0f);
            }
        }
    }
    int anyShadow = __syncthreads_or(shadowed);
    int anyLit = __syncthreads_or(lit);
    // max is exact in any order: wave64 butterfly, then across the 4 waves through LDS
    for (int o = 32; o > 0; o >>= 1)
        r = fmax2(r, __shfl_xor(r, o, 64));
    int tid = (int)threadIdx.y * 16 + (int)threadIdx.x;
    if ((tid & 63) == 0)
        smax[tid >> 6] = r;
    __syncthreads();
    if (tid == 0) {
        float m = fmax2(fmax2(smax[0], smax[1]), fmax2(smax[2], smax[3]));
        uint32_t ri = (uint32_t)__builtin_floorf(m + 0.999f);
        ri = ri > 255u ? 255u : ri;
        st<uint16_t>(p.tiles, tx, ty, 2, (uint16_t)((anyShadow ? 1u : 0u) | (anyLit ? 2u : 0u) | (ri << 8)));
    }
}

// one thread per tile
__global__ __launch_bounds__(256) void k_sigma_smooth_tiles(const SigmaParams p) {
    const FrameConsts& c = p.c;
    int total = c.tilesX * c.tilesY;
    int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= total)
        return;
    int ty = i / c.tilesX, tx = i - ty * c.tilesX;
    ty += c.tileY0;
    int tilesYAll = (c.resH + 15) / 16;
    uint32_t flags = 0, r = 0;
    for (int j = -1; j <= 1; j++)
        for (int k = -1; k <= 1; k++) {
            int x = tx + k, y = ty + j;
            if (x < 0 || x >= c.tilesX || y < 0 || y >= tilesYAll)
                continue;
            uint32_t t = ld<uint16_t>(p.tiles, x, y, 2);
            flags |= t & 3u;
            r = r > (t >> 8) ? r : (t >> 8);
        }
    st<uint16_t>(p.tilesSmooth, tx, ty, 2, (uint16_t)((flags == 3u ? 1u : 0u) | (r << 8)));
}

template <int PASS>
__global__ __launch_bounds__(256) void k_sigma_blur(const SigmaParams p) {
    const FrameConsts& c = p.c;
    int tx, ty;
    if (!xcd_tile(c, tx, ty))
        return;
    int x = tx * 16 + (int)threadIdx.x, y = ty * 16 + (int)threadIdx.y;
    if (!(x < c.W && y >= c.ownY0 && y < c.ownY1))
        return;
    const PlaneRef& inPen = PASS == 0 ? p.inPen : p.pen1;
    const PlaneRef& outSh = PASS == 0 ? p.shadow1 : p.shadow2;
    uint2 graw = ld_guide(p.guide, x, y);
    float z = u2f(graw.x);
    if (!(absf(z) <= c.denoisingRange)) {
        st<uint2>(outSh, x, y, 8, uint2{0u, 0u});
        if (PASS == 0)
            st<uint16_t>(p.pen1, x, y, 2, (uint16_t)0);
        return;
    }
    float absZ = absf(z);
    // (round 3: fetching guide, penumbra and signal texels in one batch, ahead of the range test, measured +3 % - like every other
    // attempt to trade dependent round trips for bytes in these kernels; only the tile word moved, to the scalar data path)
    float pen = h2f(ld<uint16_t>(inPen, x, y, 2));
    bool lit = PASS == 0 ? pen >= NRD_FP16_MAX : !(pen > 0.0f);
    f4 center = PASS == 0 ? input_visibility(p, x, y, pen) : unpack_h4(ld<uint2>(p.shadow1, x, y, 8));
    const uint32_t tile = ld_tile_u16(p.tilesSmooth, tx, ty);
    if (!(tile & 1u)) {
        st<uint2>(outSh, x, y, 8, pack_h4(center));
        if (PASS == 0)
            st<uint16_t>(p.pen1, x, y, 2, f2h(lit ? 0.0f : pen));
        return;
    }
    float pixelWorld = c.unproject * zpersp(absZ);
    float radiusPx = lit ? (float)(tile >> 8) : pen * rcp_(pixelWorld);
    radiusPx = fmin2(radiusPx, MAX_PIXEL_RADIUS);
    const int gy0 = y + c.yOff;
    Guide g = decode_guide(graw, c.denoisingRange);
    const float rx = fma_(c.pv[2], (float)x, c.pv[0]), ry = fma_(c.pv[3], (float)gy0, c.pv[1]); // view ray (orthographic: view-space xy)
    f3 Xv = {zpersp(z) * rx, zpersp(z) * ry, z};
    f3 Nv = rot3(c.w2v, g.n);
    float frustumSize = c.minRectDimMulUnproject * zpersp(absZ);
    float geoA = rcp_(p.planeDistanceSensitivity * frustumSize);
    float gax = Nv.x * c.pv[2] * geoA, gay = Nv.y * c.pv[3] * geoA;
    // plane distance of a tap = |zs * (ga0 + gax px + gay gy) + geoB|; orthographic: |zs * geoB + (ga0 + gax px + gay gy)|
    float ga0 = ORTHO ? (fma_(Nv.x, c.pv[0], Nv.y * c.pv[1]) - dot3(Nv, Xv)) * geoA : fma_(Nv.x, c.pv[0], fma_(Nv.y, c.pv[1], Nv.z)) * geoA;
    float geoB = ORTHO ? Nv.z * geoA : -dot3(Nv, Xv) * geoA;
    f3 T, B;
    basis3(Nv, T, B);
    float ju[4]; // the kernel basis in pixels per pixel of radius (nrd_device.h kernel_basis_px: the depth cancels, no reciprocal)
    kernel_basis_px(c, z, rx, ry, T, B, ju);
    float jtx = ju[0] * radiusPx, jty = ju[1] * radiusPx;
    float jbx = ju[2] * radiusPx, jby = ju[3] * radiusPx;
    constexpr bool PER_PIXEL = PASS == 0; // Blur rotates per pixel, PostBlur per frame (coalesced gathers)
    uint32_t h = hash_px(PER_PIXEL ? (uint32_t)x : 0u, PER_PIXEL ? (uint32_t)gy0 : 0u, c.frameIndex, 17u + (uint32_t)PASS);
    float rc = c.rot[h & 63u][0], rs = c.rot[h & 63u][1];
    { // rotation folded into the Jacobian (J . R): the taps then are the unrotated disk, 4 fma per tap
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
    if (radiusPx > 0.0f) {
        const float cx = (float)x + 0.5f, cy = (float)gy0 + 0.5f;
#pragma unroll 2
        for (int t = 0; t < 8; t++) {
            const float ox = g_poisson8[t][0], oy = g_poisson8[t][1];
            float fpx = __builtin_floorf(fma_(ox, jtx, fma_(oy, jbx, cx)));
            float fpy = __builtin_floorf(fma_(ox, jty, fma_(oy, jby, cy)));
            // loads first (clamped address), validation after: one memory round trip per tap
            bool valid = fpx >= 0.0f && fpx < (float)c.W && fpy >= 0.0f && fpy < (float)c.H;
            int px = (int)clampf(fpx, 0.0f, (float)(c.W - 1)), gy = (int)clampf(fpy, 0.0f, (float)(c.H - 1)), py = gy - c.yOff;
            int ddx = px - x, ddy = gy - gy0;
            valid = valid && !(ddx > BLUR_REACH || -ddx > BLUR_REACH || ddy > BLUR_REACH || -ddy > BLUR_REACH) && py >= 0 && py < c.resH;
            py = py < 0 ? 0 : (py >= c.resH ? c.resH - 1 : py);
            float zs = ld<float>(p.guide, px, py, GUIDE_BYTES, 0);
            uint16_t praw = ld<uint16_t>(inPen, px, py, 2);
            uint2 sraw = PASS == 0 ? uint2{0u, 0u} : ld<uint2>(p.shadow1, px, py, 8);
            if (!valid || !(absf(zs) <= c.denoisingRange))
                continue;
            float ga = fma_(gax, fpx, fma_(gay, fpy, ga0));
            float w = g_poisson8[t][2] * smoothstep01(1.0f - absf(ORTHO ? fma_(zs, geoB, ga) : fma_(zs, ga, geoB)));
            float ps = h2f(praw);
            bool lits = PASS == 0 ? ps >= NRD_FP16_MAX : !(ps > 0.0f);
            f4 sv = PASS == 0 ? input_visibility(p, px, py, ps) : unpack_h4(sraw);
            sum = fma4(sv, w, sum);
            wsum += w;
            if (!lits) {
                penSum = fma_(ps, w, penSum);
                penW += w;
            }
        }
    }
    st<uint2>(outSh, x, y, 8, pack_h4(mul4(sum, rcp_(wsum))));
    if (PASS == 0)
        st<uint16_t>(p.pen1, x, y, 2, f2h(penW > 0.0f ? penSum * rcp_(penW) : 0.0f));
}

NRD_DEV void store_out(const SigmaParams& p, int x, int y, uint32_t packed) {
    if (p.outBpt == 1)
        st<uint8_t>(p.out, x, y, 1, (uint8_t)(packed & 255u));
    else
        st<uint32_t>(p.out, x, y, 4, packed);
}

// Two kernels over the same grid, each taking the tiles of its kind and leaving at once on the others: COPY = tiles without penumbra
// in reach (most of a frame), a streaming copy that wants many waves in flight (17 VGPRs: 8 waves per SIMD); !COPY = the stabilization
// proper, which keeps 25 window texels + two footprints in registers and runs at 3 waves per SIMD. As ONE kernel the copy tiles ran at
// the register budget of the heavy path: 1.6 TB/s for a 24 B/px copy.
template <bool COPY>
__global__ __launch_bounds__(256) NRD_WAVES_PER_EU(COPY ? 8 : 2) void k_sigma_temporal_stabilization(const SigmaParams p) {
    __shared__ uint2 tile[COPY ? 1 : 400]; // RGBA16F texels, 0xffffffff marks "sky / outside"
    const FrameConsts& c = p.c;
    int tx, ty;
    if (!xcd_tile(c, tx, ty))
        return;
    const int x = tx * 16 + (int)threadIdx.x, y = ty * 16 + (int)threadIdx.y;
    const bool live = x < c.W && y >= c.ownY0 && y < c.ownY1;
    // tile check (block-uniform, scalar load): no penumbra in this tile or its 8 neighbours (SmoothTiles) -> every texel of the 5x5
    // windows is an unfiltered lit / umbra value: nothing to stabilize, the history simply follows the signal - a streaming copy
    const bool copyTile = !(ld_tile_u16(p.tilesSmooth, tx, ty) & 1u);
    if (copyTile != COPY)
        return;
    if constexpr (COPY) {
        if (!live)
            return;
        float u = ((float)x + 0.5f) * c.invW;
        bool split = u < c.splitScreen;
        float z = ld<float>(p.guide, x, y, GUIDE_BYTES, 0);
        const uint2 sh = ld<uint2>(p.shadow2, x, y, 8);
        bool sky = !(absf(z) <= c.denoisingRange);
        uint32_t packed = sky ? 0u : encode_shadow(unpack_h4(sh));
        st<uint32_t>(p.hist, x, y, 4, packed);
        store_out(p, x, y, split ? encode_shadow(input_visibility(p, x, y, h2f(ld<uint16_t>(p.inPen, x, y, 2)))) : packed);
        return;
    } else {
    // Round 3: what this kernel waits for is memory round trips (it used to make about ten dependent ones: staging guide -> staging
    // texel, twice, barrier, centre guide -> centre texel -> motion vector -> four footprint guides -> four history texels). Now three:
    // (1) the centre loads of the pixel and the ring of the 20x20 window - the 256 interior positions ARE the threads' own pixels, staged
    // from the centre loads -, all unconditional at clamped coordinates; (2) the footprint: four guide and four history texels at
    // clamped positions, validated with selects afterwards; (3) nothing - the 5x5 moments are read from LDS while (2) travels.
    const int cxp = imin(x, c.W - 1), cyp = imin(imax(y, 0), c.resH - 1); // clamped: threads outside the frame still stage
    const uint2 graw = ld_guide(p.guide, cxp, cyp);
    const uint2 craw = ld<uint2>(p.shadow2, cxp, cyp, 8);
    const uint2 mvTexel = ld<uint2>(p.inMV, cxp, cyp, 8);
    const uint32_t mixRaw = c.mixAvail ? (uint32_t)ld<uint8_t>(p.inMix, cxp, cyp, 1) : 0u;
    const int tid = (int)threadIdx.y * 16 + (int)threadIdx.x;
    int rlx = 0, rly = 0;
    const bool ringOn = ring_pos(tid, rlx, rly);
    float ringZ;
    uint2 ringT;
    bool ringIn;
    {
        const int px = tx * 16 + rlx - 2, py = ty * 16 + rly - 2, gy = py + c.yOff;
        ringIn = px >= 0 && px < c.W && gy >= 0 && gy < c.H && py >= 0 && py < c.resH;
        const int cx = imin(imax(px, 0), c.W - 1), cy = imin(imax(py, 0), c.resH - 1);
        ringZ = ld<float>(p.guide, cx, cy, GUIDE_BYTES, 0);
        ringT = ld<uint2>(p.shadow2, cx, cy, 8);
    }
    // ---- reprojection of the centre, footprint loads
    const int gy0 = y + c.yOff;
    float u = ((float)x + 0.5f) * c.invW, v = ((float)gy0 + 0.5f) * c.invH;
    bool split = u < c.splitScreen;
    float z = u2f(graw.x);
    Guide g = decode_guide(graw, c.denoisingRange);
    f3 Xv = reconstruct_px(c.pv, (float)x, (float)gy0, z);
    f4 mvRaw = unpack_h4(mvTexel);
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
    float px = fma_(su, (float)c.Wprev, -0.5f), py = fma_(sv, (float)c.Hprev, -0.5f);
    float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
    float fx = px - fx0, fy = py - fy0;
    // (a sky / outside pixel or an unusable position computes harmless clamped addresses; NaN compares false: not sane)
    const bool sane = c.historyOk && uvOk && fx0 >= -2.0f && fx0 <= (float)c.Wprev + 1.0f && fy0 >= -2.0f && fy0 <= (float)c.Hprev + 1.0f;
    const int ix = sane ? (int)fx0 : -4, iy = sane ? (int)fy0 : -4;
    uint2 fg[4];
    uint32_t fh[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int ttx = imin(imax(ix + (i & 1), 0), c.Wprev - 1), tty = imin(imax(iy + (i >> 1) - c.yOff, 0), c.resH - 1);
        fg[i] = ld_guide(p.guidePrev, ttx, tty);
        fh[i] = ld<uint32_t>(p.histPrev, ttx, tty, 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the 20x20 window: interior from the centre loads, ring from the ring loads
    {
        const bool ok = x < c.W && gy0 >= 0 && gy0 < c.H && y >= 0 && y < c.resH && absf(z) <= c.denoisingRange;
        tile[((int)threadIdx.y + 2) * 20 + (int)threadIdx.x + 2] = ok ? craw : uint2{0xffffffffu, 0xffffffffu};
    }
    if (ringOn)
        tile[rly * 20 + rlx] = (ringIn && absf(ringZ) <= c.denoisingRange) ? ringT : uint2{0xffffffffu, 0xffffffffu};
    __syncthreads();
    if (!live)
        return;
    if (!(absf(z) <= c.denoisingRange)) {
        st<uint32_t>(p.hist, x, y, 4, 0u);
        store_out(p, x, y, split ? encode_shadow(input_visibility(p, x, y, h2f(ld<uint16_t>(p.inPen, x, y, 2)))) : 0u);
        return;
    }
    f4 cur = unpack_h4(craw);
    float m1[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; j++)
#pragma unroll
        for (int i = 0; i < 5; i++) {
            uint2 raw = tile[((int)threadIdx.y + j) * 20 + (int)threadIdx.x + i];
            f4 f = raw.x == 0xffffffffu && raw.y == 0xffffffffu ? cur : unpack_h4(raw);
            m1[0] += f.x;
            m2[0] = fma_(f.x, f.x, m2[0]);
            m1[1] += f.y;
            m2[1] = fma_(f.y, f.y, m2[1]);
            m1[2] += f.z;
            m2[2] = fma_(f.z, f.z, m2[2]);
            m1[3] += f.w;
            m2[3] = fma_(f.w, f.w, m2[3]);
        }
    f4 hist = cur;
    bool have = false;
    if (sane) {
        f3 NvPrev = rot3(c.w2vPrev, g.n);
        float thrBase = c.disocclusionThreshold;
        if (c.mixAvail) // per-pixel blend toward disocclusionThresholdAlternate
            thrBase = lerpf(thrBase, c.disoccAlt, (float)mixRaw * (1.0f / 255.0f));
        float threshold = thrBase * c.minRectDimMulUnproject * zpersp(absf(XvPrev.z));
        float bw[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
        float planeRef = dot3(NvPrev, XvPrev);
        float g0 = ORTHO ? fma_(NvPrev.x, c.pvPrev[0], NvPrev.y * c.pvPrev[1]) : fma_(NvPrev.x, c.pvPrev[0], fma_(NvPrev.y, c.pvPrev[1], NvPrev.z));
        float gx = NvPrev.x * c.pvPrev[2], gyc = NvPrev.y * c.pvPrev[3];
        f4 sum = {0, 0, 0, 0};
        float wsum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int ttx = ix + (i & 1), gy = iy + (i >> 1), tty = gy - c.yOff;
            bool ok = (ttx >= 0) & (ttx < c.Wprev) & (gy >= 0) & (gy < c.Hprev) & (tty >= c.prevY0) & (tty < c.prevY1); // (bitwise: one basic block)
            Guide gp = decode_guide(fg[i], c.denoisingRange);
            float lin = fma_(gx, (float)ttx, fma_(gyc, (float)gy, g0));
            float plane = ORTHO ? fma_(gp.z, NvPrev.z, lin) : gp.z * lin;
            const bool planeOk = absf(plane - planeRef) <= threshold, normalOk = dot3(g.n, gp.n) > PREV_NORMAL_COS;
            ok = ok & !gp.sky & planeOk & normalOk;
            // a rejected texel is selected out (the history plane holds RGBA8: any bit pattern decodes to finite values, but the sums
            // must see exactly the texels the oracle adds)
            const f4 acc = fma4(decode_shadow(fh[i]), bw[i], sum);
            sum.x = ok ? acc.x : sum.x; // (component by component: a select of the whole struct goes through scratch memory)
            sum.y = ok ? acc.y : sum.y;
            sum.z = ok ? acc.z : sum.z;
            sum.w = ok ? acc.w : sum.w;
            wsum = ok ? wsum + bw[i] : wsum;
        }
        if (wsum > 0.0f) {
            hist = mul4(sum, rcp_(wsum));
            have = true;
        }
    }
    float w = have ? p.maxStab / (1.0f + p.maxStab) : 0.0f;
    float hc[4] = {hist.x, hist.y, hist.z, hist.w}, cc[4] = {cur.x, cur.y, cur.z, cur.w}, o[4];
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        float a = m1[ch] * (1.0f / 25.0f), b = m2[ch] * (1.0f / 25.0f);
        float sigma = __builtin_sqrtf(fmax2(fma_(-a, a, b), 0.0f)) * STAB_SIGMA_SCALE;
        float hcl = clampf(hc[ch], a - sigma, a + sigma);
        o[ch] = lerpf(cc[ch], hcl, w);
    }
    uint32_t packed = encode_shadow({o[0], o[1], o[2], o[3]});
    st<uint32_t>(p.hist, x, y, 4, packed);
    store_out(p, x, y, split ? encode_shadow(input_visibility(p, x, y, h2f(ld<uint16_t>(p.inPen, x, y, 2)))) : packed);
    }
}

// REFERENCE: running mean in an RGBA32F history (Source/NRDSample.cpp:4213-4224; in place, :484-485)
__global__ __launch_bounds__(256) void k_reference_accumulate(const ReferenceParams p) {
    const FrameConsts& c = p.c;
    int tx, ty;
    if (!xcd_tile(c, tx, ty))
        return;
    int x = tx * 16 + (int)threadIdx.x, y = ty * 16 + (int)threadIdx.y;
    if (!(x < c.W && y >= c.ownY0 && y < c.ownY1))
        return;
    f4 s = unpack_h4(ld<uint2>(p.in, x, y, 8));
    f4 h = s;
    if (!p.restart) {
        float4 hv = ld<float4>(p.hist, x, y, 16);
        h = lerp4({hv.x, hv.y, hv.z, hv.w}, s, p.weight);
    }
    st<float4>(p.hist, x, y, 16, float4{h.x, h.y, h.z, h.w});
    float u = ((float)x + 0.5f) * c.invW;
    st<uint2>(p.out, x, y, 8, pack_h4(u < c.splitScreen ? s : h));
}

dim3 grid_for(const FrameConsts& c) { return dim3((unsigned)xcd_grid_blocks(c.tilesX, c.tilesY), 1, 1); }

NRD_KERNELS_END

namespace NRD_PROJ_NS {

void launch_sigma_classify_tiles(const SigmaParams& p, hipStream_t s) {
    hipLaunchKernelGGL(k_sigma_classify_tiles, dim3((unsigned)p.c.tilesX, (unsigned)p.c.tilesY, 1), dim3(16, 16, 1), 0, s, p);
}
void launch_sigma_smooth_tiles(const SigmaParams& p, hipStream_t s) {
    int total = p.c.tilesX * p.c.tilesY;
    hipLaunchKernelGGL(k_sigma_smooth_tiles, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
}
void launch_sigma_blur(const SigmaParams& p, int pass, hipStream_t s) {
    if (pass == 0)
        hipLaunchKernelGGL(k_sigma_blur<0>, grid_for(p.c), dim3(16, 16, 1), 0, s, p);
    else
        hipLaunchKernelGGL(k_sigma_blur<1>, grid_for(p.c), dim3(16, 16, 1), 0, s, p);
}
void launch_sigma_temporal_stabilization(const SigmaParams& p, hipStream_t s) {
    hipLaunchKernelGGL(k_sigma_temporal_stabilization<true>, grid_for(p.c), dim3(16, 16, 1), 0, s, p);  // tiles without penumbra in reach: copy
    hipLaunchKernelGGL(k_sigma_temporal_stabilization<false>, grid_for(p.c), dim3(16, 16, 1), 0, s, p); // the others (disjoint tiles: any order)
}
void launch_reference_accumulate(const ReferenceParams& p, hipStream_t s) {
    hipLaunchKernelGGL(k_reference_accumulate, grid_for(p.c), dim3(16, 16, 1), 0, s, p);
}

} // namespace NRD_PROJ_NS

} // namespace nrdhip
