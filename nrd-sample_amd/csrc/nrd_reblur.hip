// nrd_reblur.hip - REBLUR_DIFFUSE / REBLUR_SPECULAR / REBLUR_DIFFUSE_SPECULAR passes as gfx950 HIP kernels.
//
// Replaces the REBLUR HLSL pass set of the reference's absent External/NRD submodule behind nrd::Integration::Denoise
// (Source/NRDSample.cpp:521). Pass graph per SURVEY.md 8a-5:
//   ClassifyTiles(+guide packing) -> PrePass -> TemporalAccumulation -> HistoryFix -> Blur -> PostBlur -> TemporalStabilization
// Data layout (DESIGN.md "HBM layout"): one 16-byte guide texel {viewZ f32, normal 3 x f16, roughness f16, materialID} per
// pixel so a bilateral tap costs ONE 16-byte gather and four converts for all guides; diffuse+specular radiance interleaved
// in one 16-byte texel; accumulation speeds 2 x u8 in one 16-bit texel. Workgroups are 16x16 pixel tiles; every XCD walks
// compact blocks of them (column bands that rotate every 1/8 of the frame height, nrd_device.h xcd_tile), so stencil / gather
// overlap between neighbouring tiles is served by one XCD's L2 and every XCD gets the same mix of sky and geometry. These are
// gather / stencil filters (bound by the texture addresser's gather rate and by VALU issue): no MFMA. The 5x5 moment
// stencils and the first RELAX A-trous iterations stage their tile (+ halo) in LDS; tiles without geometry (ClassifyTiles'
// Tiles mask) skip that staging.
#include "nrd_kernels.h"

// NRD_PART 0 (default): every launcher of this file except the non-SH Blur one; NRD_PART 1 (nrd_reblur_blur*.hip): only that one -
// the largest kernel family of the file compiles in its own translation unit (parallel build; round 1 also gave it its own flags).
#ifndef NRD_PART
#define NRD_PART 0
#endif

namespace nrdhip {

NRD_KERNELS_BEGIN

// taps in flight per wave in the software-pipelined tap loop of the spatial passes (k_spatial)
#ifndef NRD_PIPE_DEPTH
#define NRD_PIPE_DEPTH 8
#endif
#ifndef NRD_PIPE_DEPTH_WIDE
#define NRD_PIPE_DEPTH_WIDE 5
#endif
#ifndef NRD_SH_PRE_DEPTH // taps in flight: PrePass of the SH flavours (three gathers per tap: guide, SH0, SH1)
#define NRD_SH_PRE_DEPTH NRD_PIPE_DEPTH_WIDE
#endif
#ifndef NRD_SH_PRE_WAVES
#define NRD_SH_PRE_WAVES 3
#endif
#ifndef NRD_RELAX_SH_PRE_DEPTH // ... of RELAX's SH flavour (BASELINE config 4): 2 taps in flight fit 127 VGPRs = 4 waves per SIMD (5: 145 / 3 waves)
#define NRD_RELAX_SH_PRE_DEPTH NRD_SH_PRE_DEPTH
#endif
#ifndef NRD_RELAX_SH_PRE_WAVES
#define NRD_RELAX_SH_PRE_WAVES NRD_SH_PRE_WAVES
#endif
#ifndef NRD_TA_SH_WAVES // TemporalAccumulation of the SH / RELAX flavours: waves per SIMD the register allocator aims for (1: no bound)
#define NRD_TA_SH_WAVES 1
#endif
// Blur / PostBlur on tap texels (one 16-byte gather per tap): taps in flight and waves per SIMD
#ifndef NRD_TAP_DEPTH // (the default flavour's arccosine / exp2 polynomials hold more registers per tap: 8 taps in flight spill 16 bytes at 5 waves)
#define NRD_TAP_DEPTH (NRD_UPSTREAM_FORMULAS ? (NRD_ORTHO ? 5 : 6) : NRD_PIPE_DEPTH) // (orthographic flavour: 6 spill 20 bytes)
#endif
#ifndef NRD_PRE_DEPTH // taps in flight: REBLUR radiance PrePass
#define NRD_PRE_DEPTH 2
#endif
#ifndef NRD_POST_DEPTH // taps in flight: PostBlur on tap texels
#define NRD_POST_DEPTH (NRD_UPSTREAM_FORMULAS ? 2 : 3) // 6 waves per SIMD either way (75 / 73 VGPRs; one tap more in flight: 82 / 81, 5 waves)
#endif
#ifndef NRD_POST_WAVES // PostBlur on tap texels: waves per SIMD the register allocator aims for
#define NRD_POST_WAVES NRD_TAP_WAVES
#endif
#ifndef NRD_PRE_WAVES // PrePass: waves per SIMD
#define NRD_PRE_WAVES 4
#endif
#ifndef NRD_TA_LEAN // A/B switch of TemporalAccumulation's register diet. 1: everything that needs the world-space positions (parallax, accumulation
#define NRD_TA_LEAN 0 // limit, virtual position) is reduced to scalars BEFORE the footprint's 20 loads go out; 3: ... and the fused kernel does not
#endif               // carry the view position / normal / view vector across the PrePass tap loop (TemporalAccumulation derives them again)
#ifndef NRD_FUSED_SEQ_FOOTPRINTS // the fused kernel fetches its two history footprints one after the other (ta_pixel SEQ_FOOT)
#define NRD_FUSED_SEQ_FOOTPRINTS 1
#endif
#ifndef NRD_FUSED_DEPTH // taps in flight: the PrePass half of the fused PrePass + TemporalAccumulation kernel. With both footprints in one round trip
#define NRD_FUSED_DEPTH 2 // the reprojection half set the register count (115-119 VGPRs at any depth, 4 waves; depth 5 fastest); with sequential
#endif                    // footprints depth 2 fits 93 VGPRs = 5 waves (depth 3: 20 bytes of scratch)
#ifndef NRD_TAP_WAVES // 104 VGPRs at 4 waves; capped at 102 for a fifth wave: Blur 0.202 -> 0.198 ms, PostBlur 0.179 -> 0.1705
#define NRD_TAP_WAVES 5  // (depth 6 / 8 at 5 waves: equal; depth 12 / 16 at 4 waves: slower - profiles/r03_ab_tap_texels.txt)
#endif

constexpr float MAX_ACCUM = 63.0f;
constexpr float MIN_CONVERGED_RADIUS_SCALE = 0.25f;
constexpr float POST_BLUR_RADIUS_SCALE = 2.0f;
constexpr float NORMAL_ANGLE_MIN = 0.02f;
constexpr float PREV_NORMAL_COS = 0.7f;

NRD_DEV void unpack_data1(uint32_t v, float& diffA, float& specA) {
    diffA = (float)(v & 0xffu) * 0.25f;
    specA = (float)(v >> 8) * 0.25f;
}
NRD_DEV uint16_t pack_data1(float diffA, float specA) {
    uint32_t d = (uint32_t)__builtin_floorf(fma_(clampf(diffA, 0.0f, MAX_ACCUM), 4.0f, 0.5f));
    uint32_t s = (uint32_t)__builtin_floorf(fma_(clampf(specA, 0.0f, MAX_ACCUM), 4.0f, 0.5f));
    return (uint16_t)(d | (s << 8));
}

// OCCLUSION variants: the signal is the normalised hit distance alone (R16_UNORM / R16F plane, Source/NRDSample.cpp:488-501);
// internally it travels as {h, 0, 0, h} so every luma-based stage works on it unchanged
NRD_DEV f4 load_signal(const ReblurParams& p, const PlaneRef& P, int x, int y, int bpt, int off, bool occlusion) {
    if (!occlusion)
        return unpack_h4(ld<uint2>(P, x, y, bpt, off));
    uint16_t raw = ld<uint16_t>(P, x, y, 2);
    float h = p.ioF16 ? h2f(raw) : (float)raw * (1.0f / 65535.0f);
    return {h, 0.0f, 0.0f, h};
}
// DIRECTIONAL_OCCLUSION split-screen passthrough: the noisy {direction * h, h} texel rebuilt from its prepared SH0 / SH1 halves
NRD_DEV f4 dir_pass(const ReblurParams& p, int x, int y) {
    f4 a = unpack_h4(ld<uint2>(p.inDiff, x, y, 8)), b = unpack_h4(ld<uint2>(p.inDiff1, x, y, 8));
    return {b.x, b.y, b.z, a.x};
}
// OUT_DIFF_DIRECTION_HITDIST texel in the bound format
NRD_DEV uint2 pack_dir(const ReblurParams& p, f4 v) { return p.dirOutSnorm ? pack_sn4(v) : pack_h4(v); }
// decode of a gathered signal texel (the tap loops gather the raw words first and decode them when the tap is consumed)
NRD_DEV f4 decode_signal(const ReblurParams& p, uint2 raw, bool occlusion) {
    if (!occlusion)
        return unpack_h4(raw);
    float h = p.ioF16 ? h2f((uint16_t)raw.x) : (float)raw.x * (1.0f / 65535.0f);
    return {h, 0.0f, 0.0f, h};
}
NRD_DEV void store_signal(const ReblurParams& p, const PlaneRef& P, int x, int y, f4 v) {
    if (!p.occlusion) {
        st<uint2>(P, x, y, 8, pack_h4(v));
        return;
    }
    st<uint16_t>(P, x, y, 2, p.ioF16 ? f2h(v.x) : (uint16_t)__builtin_floorf(fma_(sat(v.x), 65535.0f, 0.5f)));
}

// whole radiance texel (8 / 16 / 32 bytes: one or two signals, SH0 [+ SH1]) in as few loads as its size allows
template <int BYTES>
NRD_DEV void load_texel(const PlaneRef& P, int x, int y, uint2 (&t)[BYTES / 8]) {
    if constexpr (BYTES == 8) {
        t[0] = ld<uint2>(P, x, y, 8);
    } else {
#pragma unroll
        for (int k = 0; k < BYTES / 16; k++) {
            uint4 v = ld<uint4>(P, x, y, BYTES, k * 16);
            t[2 * k] = uint2{v.x, v.y};
            t[2 * k + 1] = uint2{v.z, v.w};
        }
    }
}

// ... and out in as few stores: the signals of a pixel share one texel, so their results leave together (a memory instruction costs the
// texture addresser the same 16 cycles per wave whether it carries 8 or 16 bytes per lane)
template <int BYTES, bool STREAM = false>
NRD_DEV void store_texel(const PlaneRef& P, int x, int y, const uint2 (&t)[BYTES / 8]) {
    if constexpr (BYTES == 8) {
        if (STREAM)
            st_stream<uint2>(P, x, y, 8, t[0]);
        else
            st<uint2>(P, x, y, 8, t[0]);
    } else {
#pragma unroll
        for (int k = 0; k < BYTES / 16; k++) {
            const uint4 v = {t[2 * k].x, t[2 * k].y, t[2 * k + 1].x, t[2 * k + 1].y};
            if (STREAM)
                st_stream<uint4>(P, x, y, BYTES, v, k * 16);
            else
                st<uint4>(P, x, y, BYTES, v, k * 16);
        }
    }
}

// REBLUR / RELAX::Tiles (written by ClassifyTiles): 1 = no pixel of the 16x16 tile has geometry. One byte per tile, same address for
// the whole workgroup: a scalar value
// (`tflag`: the flag as xcd_tile delivered it with the tile - launches that carry the flags in table order - or -1: read the plane)
NRD_DEV bool tile_is_sky(const PlaneRef& tiles, int tx, int ty, int tflag) { return (tflag >= 0 ? (uint32_t)tflag : ld_tile_u8(tiles, tx, ty)) != 0u; }
NRD_DEV bool tile_is_sky(const ReblurParams& p, int tx, int ty, int tflag) { return tile_is_sky(p.tiles, tx, ty, tflag); }

// pixel of this thread inside its XCD-swizzled tile; false = nothing to do
NRD_DEV bool my_pixel(const FrameConsts& c, int& x, int& y, int& tx, int& ty, int& tflag, const bool pin = true) {
    if (!xcd_tile(c, tx, ty, tflag, pin))
        return false;
    x = tx * 16 + (int)threadIdx.x;
    y = ty * 16 + (int)threadIdx.y;
    return x < c.W && y >= c.ownY0 && y < c.ownY1;
}
NRD_DEV bool my_pixel(const FrameConsts& c, int& x, int& y, int& tx, int& ty) {
    int tflag;
    return my_pixel(c, x, y, tx, ty, tflag);
}

// TemporalAccumulation runs one WAVE per workgroup (16 x 4 pixels, a quarter of a tile; the four quarters of a tile are consecutive
// workgroups of one XCD): a finished wave is replaced at once instead of waiting for the other three of a 256-thread workgroup
// (-4 % on that kernel; its footprint gathers follow the motion vectors and share little inside a tile anyway).
#ifndef NRD_WG64
#define NRD_WG64 1
#endif
NRD_DEV bool my_pixel_w(const FrameConsts& c, int& x, int& y, int& tx, int& ty, int& tflag) {
#if NRD_WG64
    const int b = (int)blockIdx.x, jj = b >> 3;
    if (!xcd_tile_of(c, b & 7, jj >> 2, tx, ty, tflag))
        return false;
    x = tx * 16 + (int)threadIdx.x;
    y = ty * 16 + (jj & 3) * 4 + (int)threadIdx.y;
    return x < c.W && y >= c.ownY0 && y < c.ownY1;
#else
    return my_pixel(c, x, y, tx, ty, tflag);
#endif
}
NRD_DEV bool my_pixel_w(const FrameConsts& c, int& x, int& y, int& tx, int& ty) {
    int tflag;
    return my_pixel_w(c, x, y, tx, ty, tflag);
}

// =====================================================================================================================
// K0 ClassifyTiles + guide packing
// =====================================================================================================================
#ifndef NRD_SKIP_SKY_TILES // 1: PrePass, TemporalAccumulation and PostBlur write nothing in tiles without geometry
#define NRD_SKIP_SKY_TILES 1
#endif
#ifndef NRD_CT_TILES // tiles per ClassifyTiles workgroup (a horizontal run)
#define NRD_CT_TILES 4
#endif
// ---- roughness-only terms of the specular kernel set-up -------------------------------------------------------------------------
// Two exp2 polynomials, two software square roots, an arctangent: ~90 instructions that hang on the 10-bit roughness code of the pixel
// alone, evaluated in the PrePass, Blur and PostBlur of every specular pixel. Round 6: ClassifyTiles tabulates them per frame (1024
// codes, the same functions) and a pixel fetches its entry with ONE 16-byte load issued as soon as its guide texel is in, consumed by the
// specular signal's set-up BEHIND the geometry and the diffuse signal's set-up (a sched_barrier holds that order: round 5 measured
// the table with the load in front of its consumer and found the round trip cost what the instructions saved - profiles/r05_ab_roughness_table.txt)
#ifndef NRD_ROUGH_LUT
#define NRD_ROUGH_LUT 1
#endif
struct RoughTerms {
    float df, smc, angle0, hitFactor; // spec_dominant_factor, spec_magic_curve, spec_lobe_half_angle, reblur_hitdist_factor
};
NRD_DEV RoughTerms rough_terms(const float* hp, const float rough) {
    return {spec_dominant_factor(rough), spec_magic_curve(rough), spec_lobe_half_angle(rough), reblur_hitdist_factor(hp, rough)};
}
// a pixel's entry: issued by the caller right behind its guide texel (`code` = the low 10 bits of the guide's depth word)
NRD_DEV uint4 rough_terms_load(const float* lut, const uint32_t code) { return *reinterpret_cast<const uint4*>(lut + (code & 1023u) * 4u); }
NRD_DEV RoughTerms rough_terms_of(const uint4 raw) { return {u2f(raw.x), u2f(raw.y), u2f(raw.z), u2f(raw.w)}; }
// the dominant factor alone (the reprojection passes: virtual motion, its blend amount): one dword of the pixel's entry - the guide's depth
// word read as a float IS the depth, so the code is still in its low bits
NRD_DEV float dominant_factor_of(const ReblurParams& p, const Guide& g) {
    return NRD_ROUGH_LUT ? p.roughLut[(f2u(g.z) & 1023u) * 4u] : spec_dominant_factor(g.roughness);
}
// second table behind the first (floats 4096 ..): the roughness-only terms of the specular accumulation limit (spec_accum_limit) -
// {1 - 2^(-200 r^2), log2(r)}: an exp2 and a log2 polynomial, ~47 instructions per pixel of TemporalAccumulation
constexpr uint32_t ROUGH_LUT_ACCUM = 4096u;
static_assert(NRDHIP_ROUGH_LUT_FLOATS == 4096 + 2048, "nrd_kernels.h: the allocation holds both tables");
NRD_DEV float2 accum_terms_of(const ReblurParams& p, const Guide& g) { return *reinterpret_cast<const float2*>(p.roughLut + ROUGH_LUT_ACCUM + (f2u(g.z) & 1023u) * 2u); }
__global__ __launch_bounds__(256) void k_classify_tiles(const ReblurParams p) {
    const FrameConsts& c = p.c;
    // a streaming pass without neighbour reads: plain 2-D grid (the XCD traversal of the other passes costs a wave ~700 cycles of
    // scalar index arithmetic and buys this pass nothing: 0.0485 -> 0.0462 ms, profiles/r03_ab_setup_planes.txt). What it waits for is
    // the one round trip of its two loads, so a workgroup takes a run of NRD_CT_TILES horizontally adjacent tiles - every thread has the
    // loads of 4 pixels in flight before the first returns - and a run of 4 tiles is two whole 128-byte lines of the 4-byte input
    // planes per row: no line is fetched by the L2s of two XCDs (what the tile pairing of the one-tile version was for)
    __shared__ int sGeo[NRD_CT_TILES];
    const int tid = (int)threadIdx.y * 16 + (int)threadIdx.x;
    if (tid < NRD_CT_TILES)
        sGeo[tid] = 0;
    __syncthreads();
    const int tx0 = (int)blockIdx.x * NRD_CT_TILES, ty = (int)blockIdx.y + c.tileY0;
    const int y = ty * 16 + (int)threadIdx.y;
    const bool rowValid = y < c.resH && (y + c.yOff) < c.H && (y + c.yOff) >= 0;
    float z[NRD_CT_TILES];
    uint32_t nr[NRD_CT_TILES];
#pragma unroll
    for (int k = 0; k < NRD_CT_TILES; k++) {
        const int x = (tx0 + k) * 16 + (int)threadIdx.x;
        const bool valid = rowValid && x < c.W;
        z[k] = valid ? ld<float>(p.inZ, x, y, 4) : 0.0f;
        nr[k] = valid ? ld<uint32_t>(p.inNR, x, y, 4) : 0u;
    }
#pragma unroll
    for (int k = 0; k < NRD_CT_TILES; k++) {
        const int x = (tx0 + k) * 16 + (int)threadIdx.x;
        if (rowValid && x < c.W) {
            const float zs = z[k] * c.viewZScale;
            bool geo;
            st<uint2>(p.guide, x, y, GUIDE_BYTES, encode_guide(zs, nr[k], c.denoisingRange, geo));
            if (geo)
                sGeo[k] = 1; // same value from every writer
        }
    }
    __syncthreads();
    if (tid < NRD_CT_TILES && tx0 + tid < c.tilesX) {
        st<uint8_t>(p.tiles, tx0 + tid, ty, 1, sGeo[tid] ? 0 : 1);
        // ... and once more where the workgroup that will work on this tile finds it next to its table entry (FrameConsts::tileFlags; only
        // when the other passes' grid is this grid: nrdhip.cpp tile_flags)
        if (p.tileFlagsOut)
            p.tileFlagsOut[p.tileInv[(uint32_t)(ty - c.tileY0) * (uint32_t)c.tilesX + (uint32_t)(tx0 + tid)]] = sGeo[tid] ? 0 : 1;
    }
    // the roughness table of this frame's settings (ReblurParams::roughLut): 4 codes per thread of the first workgroup
    if (NRD_ROUGH_LUT && p.roughLut && blockIdx.x == 0 && blockIdx.y == 0) {
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            const uint32_t code = (uint32_t)(tid + 256 * k);
            const RoughTerms t = rough_terms(p.hp, (float)code * (1.0f / 1023.0f)); // (the roughness decode_guide gives a pixel of this code)
            *reinterpret_cast<float4*>(p.roughLut + code * 4u) = float4{t.df, t.smc, t.angle0, t.hitFactor};
            const float r = (float)code * (1.0f / 1023.0f);
            *reinterpret_cast<float2*>(p.roughLut + ROUGH_LUT_ACCUM + code * 2u) = float2{1.0f - exp2_poly(-200.0f * r * r), log2_poly(sat(r))};
        }
    }
}

// =====================================================================================================================
// K1 PrepareInputs - recorded only when the inputs are checkerboarded or hit distances must be reconstructed (the sample's
// default operating mode, Source/NRDSample.cpp:267, :545-548). Writes dense RGBA16F copies of the noisy inputs for the PrePass:
//  * checkerboard resolve: half-width inputs hold pixel (x, y) at texel (x >> 1, y) on the squares of
//    Sequence::CheckerBoard(pixelPos, frameIndex) that carry the signal (Shaders/TraceOpaque.cs.hlsl:482-508); a pixel of the
//    other colour takes the depth-weighted mean of its left / right neighbours;
//  * hit distance reconstruction (AREA_3X3 / AREA_5X5): a texel without hit distance (w == 0) takes the bilateral mean of the
//    valid hit distances around it.
// An optional mode, written for clarity rather than peak rate (branchy 3x3 / 5x5 loop on the pixels that need it only).
// =====================================================================================================================
NRD_DEV bool has_data(int phase, int x, int gy, uint32_t frameIndex) { return phase == 2 || ((((uint32_t)x ^ (uint32_t)gy) ^ frameIndex) & 1u) == (uint32_t)phase; }

template <bool HAS_DIFF, bool HAS_SPEC>
__global__ __launch_bounds__(256) void k_prepare_inputs(const ReblurParams p) {
    constexpr int NSIG = (HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0);
    constexpr int SIG_SPEC = HAS_DIFF ? 1 : 0;
    const FrameConsts& c = p.c;
    int x, y, tx, ty;
    if (!my_pixel(c, x, y, tx, ty))
        return;
    const bool occ = p.occlusion != 0, checker = p.checker != 0, sh1 = p.prepSh1 != 0, dirOcc = p.dirOcc != 0;
    Guide g = decode_guide(ld_guide(p.guide, x, y), c.denoisingRange);
    const int gy0 = y + c.yOff;
#pragma unroll
    for (int sig = 0; sig < NSIG; sig++) {
        const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
        const PlaneRef& in = isSpec ? p.rawSpec : p.rawDiff;
        const PlaneRef& in1 = isSpec ? p.rawSpec1 : p.rawDiff1;
        const PlaneRef& out = isSpec ? p.inSpec : p.inDiff;
        const PlaneRef& out1 = isSpec ? p.inSpec1 : p.inDiff1;
        if (g.sky) {
            st<uint2>(out, x, y, 8, uint2{0u, 0u});
            if (sh1)
                st<uint2>(out1, x, y, 8, uint2{0u, 0u});
            continue;
        }
        const int phase = isSpec ? p.phaseSpec : p.phaseDiff;
        // one input position -> (SH0-like signal, SH1 texel)
        auto load_pair = [&](int sx, int sy, f4& a, f4& b) {
            if (dirOcc) { // {direction * h, h}
                uint2 tr = ld<uint2>(in, sx, sy, 8);
                f4 t = p.dirInSnorm ? unpack_sn4(tr) : unpack_h4(tr);
                a = {t.w, 0.0f, 0.0f, t.w};
                b = {t.x, t.y, t.z, 0.0f};
            } else {
                a = load_signal(p, in, sx, sy, 8, 0, occ);
                b = sh1 ? unpack_h4(ld<uint2>(in1, sx, sy, 8)) : f4{0, 0, 0, 0};
            }
        };
        f4 v = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
        if (has_data(phase, x, gy0, c.frameIndex)) {
            int sx = checker ? x >> 1 : x;
            load_pair(sx, y, v, v1);
        } else { // checkerboard resolve from the left / right neighbours (they carry this signal)
            float invDz = rcp_(0.03f * fmax2(absf(g.z), 1e-6f));
            float wn[2];
            bool ok[2];
            f4 vn[2], v1n[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int n = 0; n < 2; n++) {
                int px = x + (n ? 1 : -1);
                ok[n] = px >= 0 && px < c.W;
                int cpx = imin(imax(px, 0), c.W - 1);
                Guide gn = decode_guide(ld_guide(p.guide, cpx, y), c.denoisingRange);
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
        if (p.reconRadius > 0 && v.w == 0.0f) { // no hit distance: reconstruct it from the neighbourhood
            PixelGeo pg = pixel_geo(c, g, x, gy0, p.planeDistanceSensitivity);
            float rough = isSpec ? g.roughness : 1.0f;
            uint32_t minMat = isSpec ? p.minMatSpec : p.minMatDiff;
            float angle = spec_lobe_half_angle(rough) * p.lobeAngleFraction;
            float normalW = rcp_(fmax2(angle, NORMAL_ANGLE_MIN));
            normalW *= strand_normal_relax(c, g.mat, absf(g.z)); // CommonSettings::strandMaterialID: thin strands relax the normal test
            float normalW2 = nw_param(normalW);
            float roughA = rcp_(lerpf(0.01f, 1.0f, sat(rough * p.roughnessFraction)));
            float roughB = -rough * roughA;
            float sum = 0.0f, wsum = 0.0f;
            for (int j = -p.reconRadius; j <= p.reconRadius; j++)
                for (int i = -p.reconRadius; i <= p.reconRadius; i++) {
                    if (i == 0 && j == 0)
                        continue;
                    int px = x + i, py = y + j, gy = py + c.yOff;
                    if (px < 0 || px >= c.W || gy < 0 || gy >= c.H || py < 0 || py >= c.resH)
                        continue;
                    if (!has_data(phase, px, gy, c.frameIndex))
                        continue;
                    Guide gs = decode_guide(ld_guide(p.guide, px, py), c.denoisingRange);
                    if (gs.sky || material_mismatch(g.mat, gs.mat, minMat))
                        continue;
                    f4 hv, hv1;
                    load_pair(checker ? px >> 1 : px, py, hv, hv1);
                    float h = hv.w;
                    if (!(h > 0.0f))
                        continue;
                    float w = geo_weight(pg, (float)px, (float)gy, gs.z);
                    w *= normal_weight(normal_dist2(normal_codes(g.nw), gs.nw), normalW2);
                    if (isSpec)
                        w *= smoothstep01(1.0f - absf(fma_(gs.roughness, roughA, roughB)));
                    sum = fma_(h, w, sum);
                    wsum += w;
                }
            if (wsum > 0.0f)
                v.w = sum * (rcp_(wsum));
        }
        if (occ || dirOcc)
            v = {v.w, 0.0f, 0.0f, v.w};
        st<uint2>(out, x, y, 8, pack_h4(v));
        if (sh1)
            st<uint2>(out1, x, y, 8, pack_h4(v1));
    }
}

// The sample's DEFAULT operating point (Source/NRDSample.cpp:267, :545-548: RESOLUTION_HALF tracing, CheckerboardMode::WHITE, no hit distance
// reconstruction) as its own kernel: radiance signals of a diffuse + specular denoiser, each on one colour of the checkerboard. Every pixel
// then does the same thing - it HAS one signal and resolves the other from its left / right neighbours, which carry it - so the work is
// written without the two-way divergence of the general kernel above (where every wave ran "has data" and "resolve" for both signals: the
// lanes of a row alternate): three guide texels (x - 1, x, x + 1) and three signal texels, the plane of each picked per lane, all six
// loads in flight before the first is used; a plain grid of 4-tile runs like ClassifyTiles (no neighbour reuse across tiles worth an XCD
// traversal, ~700 cycles of scalar prologue less per wave). The arithmetic is the general kernel's, expression by expression.
__global__ __launch_bounds__(256) void k_prepare_checker(const ReblurParams p) {
    const FrameConsts& c = p.c;
    const int ty = (int)blockIdx.y + c.tileY0;
    const int y = ty * 16 + (int)threadIdx.y;
    if (y < c.ownY0 || y >= c.ownY1)
        return;
    const int gy0 = y + c.yOff;
#pragma unroll
    for (int k = 0; k < NRD_CT_TILES; k++) {
        const int x = ((int)blockIdx.x * NRD_CT_TILES + k) * 16 + (int)threadIdx.x;
        if (x >= c.W)
            continue;
        // the pixel carries the signal whose phase matches its colour; its neighbours carry the other one
        const bool ownIsDiff = ((((uint32_t)x ^ (uint32_t)gy0) ^ c.frameIndex) & 1u) == (uint32_t)p.phaseDiff;
        const PlaneRef& ownIn = ownIsDiff ? p.rawDiff : p.rawSpec;
        const PlaneRef& otherIn = ownIsDiff ? p.rawSpec : p.rawDiff;
        const int xl = imax(x - 1, 0), xr = imin(x + 1, c.W - 1);
        const uint2 g0 = ld_guide(p.guide, x, y), gl = ld_guide(p.guide, xl, y), gr = ld_guide(p.guide, xr, y);
#ifndef NRD_CHECKER_UNIFORM_LOADS
#define NRD_CHECKER_UNIFORM_LOADS 1
#endif
        uint2 sOwn, sL, sR;
        if (NRD_CHECKER_UNIFORM_LOADS) {
            // texel x >> 1 of BOTH input planes (one is the pixel's own signal, the other its right neighbour's for an even x - xr >> 1 == x >> 1 -
            // or its left neighbour's for an odd one), each from ONE plane for the whole wave; only the third texel - the neighbour on the far
            // side - comes from a plane picked per lane
            const uint2 d0 = ld<uint2>(p.rawDiff, x >> 1, y, 8), s0 = ld<uint2>(p.rawSpec, x >> 1, y, 8);
            const bool even = (x & 1) == 0;
            const uint2 far = ld<uint2>(otherIn, (even ? xl : xr) >> 1, y, 8);
            sOwn = ownIsDiff ? d0 : s0;
            const uint2 near = ownIsDiff ? s0 : d0;
            sL = even ? far : near;
            sR = even ? near : far;
        } else {
            sOwn = ld<uint2>(ownIn, x >> 1, y, 8);
            sL = ld<uint2>(otherIn, xl >> 1, y, 8);
            sR = ld<uint2>(otherIn, xr >> 1, y, 8);
        }
        const Guide g = decode_guide(g0, c.denoisingRange);
        // The two results leave as ONE store per output plane, the value picked per lane (own / resolved signal): every store instruction
        // then writes whole rows of one plane. Picking the PLANE per lane instead had each instruction write every other texel of two
        // planes - half-filled lines twice (NRD_CHECKER_PLANE_STORES = 1: that form, kept for the A/B of profiles/r05_ab_prepare_checker.txt)
        uint2 wOwn = {0u, 0u}, wOther = {0u, 0u};
        if (!g.sky) {
            const f4 v = unpack_h4(sOwn);
            // checkerboard resolve of the other signal (k_prepare_inputs, the same expressions)
            const float invDz = rcp_(0.03f * fmax2(absf(g.z), 1e-6f));
            float wn[2];
            bool ok[2];
            const f4 vn[2] = {unpack_h4(sL), unpack_h4(sR)};
            const uint2 gn2[2] = {gl, gr};
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const int px = x + (n ? 1 : -1);
                const Guide gn = decode_guide(gn2[n], c.denoisingRange);
                ok[n] = px >= 0 && px < c.W && !gn.sky;
                const float w = smoothstep01(1.0f - absf(gn.z - g.z) * invDz);
                wn[n] = ok[n] ? w : 0.0f;
            }
            if (!(wn[0] + wn[1] > 0.0f)) { // depth edge on both sides: plain mean of whatever exists
                wn[0] = ok[0] ? 1.0f : 0.0f;
                wn[1] = ok[1] ? 1.0f : 0.0f;
            }
            const float wsum = wn[0] + wn[1];
            f4 acc = wn[0] > 0.0f ? mul4(vn[0], wn[0]) : f4{0, 0, 0, 0};
            acc = wn[1] > 0.0f ? fma4(vn[1], wn[1], acc) : acc;
            const float inv = rcp_(wsum);
            const f4 vo = wsum > 0.0f ? mul4(acc, inv) : f4{0, 0, 0, 0};
            wOwn = pack_h4(v);
            wOther = pack_h4(vo);
        }
#ifndef NRD_CHECKER_PLANE_STORES
#define NRD_CHECKER_PLANE_STORES 0
#endif
        if (NRD_CHECKER_PLANE_STORES) {
            st<uint2>(ownIsDiff ? p.inDiff : p.inSpec, x, y, 8, wOwn);
            st<uint2>(ownIsDiff ? p.inSpec : p.inDiff, x, y, 8, wOther);
        } else {
            st<uint2>(p.inDiff, x, y, 8, uint2{ownIsDiff ? wOwn.x : wOther.x, ownIsDiff ? wOwn.y : wOther.y});
            st<uint2>(p.inSpec, x, y, 8, uint2{ownIsDiff ? wOther.x : wOwn.x, ownIsDiff ? wOther.y : wOwn.y});
        }
    }
}

// =====================================================================================================================
// Spatial filter: PrePass (VARIANT 0), Blur (1), PostBlur (2)
// =====================================================================================================================
// MODE: 0 = REBLUR radiance, 1 = RELAX radiance, 2 = OCCLUSION (hit distance only), 3 = REBLUR SH, 4 = RELAX SH (compile-time so
// the unrolled tap loop stays one basic block)
// the per-pixel body of TemporalAccumulation (defined with its kernel below): `ctex` = the pixel's PrePass result as it would sit in
// Tmp1 (packed fp16 words), `hitDist` = the tracked specular hit distance as it would sit in the hit tracker
// (`geo`: the pixel's view-space position, normal and view vector when the caller has them already - the fused kernel's PrePass half)
struct TaGeo {
    f3 Xv, Nv, V;
    float roughA; // 1 / roughness tolerance of the specular signal (KernelUnit::roughA; the default flavour's rcp_ sequence), < 0: not available
};
template <bool HAS_DIFF, bool HAS_SPEC, bool SH, bool RELAX, bool SEQ_FOOT = false>
NRD_DEV void ta_pixel(const ReblurParams& p, int x, int y, const Guide& g, const uint2 (&ctex)[((SH ? 16 : 8) * ((HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0))) / 8], float hitDist, const TaGeo* geo = nullptr);
template <bool HAS_DIFF, bool HAS_SPEC, bool SH, bool RELAX>
NRD_DEV void ta_sky_stores(const ReblurParams& p, int x, int y);

// ---- kernel set-up of a signal: the part that hangs on the pixel's geometry and roughness only, not on the pass -------------------------
// Round 6: restated - the kernel basis in pixels per pixel of radius without the perspective divide (nrd_device.h
// kernel_basis_px), explicit fma in the cross products / mirror directions, the dominant direction without the mirror vector R in
// between: ~60 instructions less per signal and pass, +1.9 % on the frame. Handing this part from the PrePass to Blur / PostBlur through
// planes (24 bytes per pixel as fp16: -290 instructions in each of the two passes) was built and measured SLOWER: Blur +9 %, the
// PrePass + TemporalAccumulation kernel +8 % for its 24 bytes of stores, PostBlur -3 %, frame -3.3 % (profiles/
// r06_ab_kernel_setup_planes.txt; the code lives in tools/variants/kernel_setup_planes.patch): in these kernels a byte per pixel costs
// what ~10 instructions cost, and a cold load at the head of a wave more.
struct KernelUnit {
    float j[4];                           // pixel offsets of the kernel's tangent / bitangent per pixel of blur radius (kernel_basis_px)
    float smc, angle0, roughA, hitFactor; // GetSpecMagicCurve(roughness), lobe half angle, 1 / roughness tolerance, hit distance factor
};
template <bool IS_SPEC>
NRD_DEV KernelUnit kernel_unit(const ReblurParams& p, const PixelGeo& pg, const float z, const f3 V, const float rough, const RoughTerms& rt) {
    KernelUnit k;
    f3 T, B;
    basis3(pg.Nv, T, B);
    if (IS_SPEC) {
        const float NoV = dot3(pg.Nv, V);
        const float df = rt.df;
        // dominant direction D = normalize(N + (R - N) df), R = 2 NoV N - V the mirror direction: N (1 + (2 NoV - 1) df) - V df
        const float alpha = fma_(fma_(NoV, 2.0f, -1.0f), df, 1.0f);
        const f3 D = normalize3({fma_(pg.Nv.x, alpha, -(V.x * df)), fma_(pg.Nv.y, alpha, -(V.y * df)), fma_(pg.Nv.z, alpha, -(V.z * df))});
        const float NoD = dot3(pg.Nv, D);
        if (NoD < 0.999f && rough < 0.95f) {
            const float n2 = 2.0f * NoD;
            const f3 Dr = {fma_(pg.Nv.x, n2, -D.x), fma_(pg.Nv.y, n2, -D.y), fma_(pg.Nv.z, n2, -D.z)}; // D mirrored at N
            T = normalize3(cross3(D, pg.Nv)); // == cross(N, Dr): the N x N term vanishes
            B = cross3(Dr, T);
            T = mul3(T, lerpf(fma_(rough, 0.5f, 0.5f), 1.0f, NoD)); // skew toward the view direction at grazing angles
        }
        k.smc = rt.smc;
        k.angle0 = rt.angle0;
        k.roughA = wrcp_(lerpf(0.01f, 1.0f, sat(rough * p.roughnessFraction))); // (weight-class: the hwt flavour's v_rcp_f32)
        k.hitFactor = rt.hitFactor;
    } else {
        k.smc = 1.0f;
        k.angle0 = spec_lobe_half_angle(1.0f);
        k.roughA = 0.0f;
        k.hitFactor = p.hitFactorDiff;
    }
    kernel_basis_px(p.c, z, pg.rx, pg.ry, T, B, k.j);
    return k;
}

// FUSED (PrePass of the REBLUR radiance flavours only): the pixel goes straight on into TemporalAccumulation - that pass reads the
// PrePass result at its OWN pixel only (Tmp1, hit tracker), so the result stays in registers: no Tmp1 store + load (32 B/px at two
// signals), one guide load, one launch and one wave prologue less, and the reprojection's memory round trips overlap with the tap
// arithmetic of the other waves of the SIMD (profiles/r04_valu_issue.txt: the spatial passes sit on VALU issue,
// TemporalAccumulation on latency). The values that cross from one half to the other are rounded exactly as the planes would have
// rounded them (packed fp16 words), so the fused dispatch is bit-identical to the two separate ones.
template <int VARIANT, int MODE, bool HAS_DIFF, bool HAS_SPEC, bool FUSED>
NRD_DEV void spatial_pixel(const ReblurParams& p, const int x, const int y, const int tx, const int ty, const int tflag) {
    static_assert(!FUSED || (VARIANT == 0 && MODE == 0), "only the REBLUR radiance PrePass continues into TemporalAccumulation");
    constexpr int NSIG = (HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0);
    constexpr bool SH = MODE == 3 || MODE == 4;
    constexpr int sb = SH ? 16 : 8;   // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    constexpr int RBPT = sb * NSIG;  // bytes per texel of the internal radiance planes
    constexpr int SIG_SPEC = HAS_DIFF ? 1 : 0;
    const FrameConsts& c = p.c;
    const PlaneRef& inP = VARIANT == 1 ? p.tmp1 : p.tmp2; // Blur reads Tmp1, PostBlur reads Tmp2 (PrePass reads the input slots)
    const PlaneRef& outP = VARIANT == 0 ? p.tmp1 : (VARIANT == 1 ? p.tmp2 : p.hist);
    const int reach = VARIANT == 0 ? p.reachPre : (VARIANT == 1 ? p.reachBlur : p.reachPost);
    constexpr bool relaxIn = VARIANT == 0 && (MODE == 1 || MODE == 4); // RELAX inputs: linear RGB + world-space hit distance
    constexpr bool occIn = VARIANT == 0 && MODE == 2;

    // REBLUR radiance Blur / PostBlur run on tap texels (nrd_device.h): guide part and signal of a pixel in one 16-byte texel per signal
    // (Blur: tapA -> tapB, PostBlur: tapB -> History) - one gather per tap, and the centre's guide comes from its own texel
    constexpr bool TAP = VARIANT != 0 && MODE == 0;
    const PlaneRef* tapIn = VARIANT == 1 ? p.tapA : p.tapB;
    // A tile without geometry (ClassifyTiles: Tiles = 1, one scalar load per workgroup). Nobody reads what PrePass or PostBlur would
    // write there: Tmp1 / the hit tracker are read at the thread's own pixel only, by passes that skip the tile as well, and the
    // history is read through footprints and 5x5 windows that test the guide first and select (or weigh with an exact 0 a plane
    // that only ever holds finite values). So they write nothing - a pixel beyond the denoising range used to cost these two passes
    // 34 + 48 bytes of traffic, a third of what a pixel with geometry costs. Blur must keep the tap texels of the tile current (PostBlur's
    // taps take their sky test from the guide inside the texel): it copies the guide texel in, without reading HistoryFix's texels.
    if (NRD_SKIP_SKY_TILES && (VARIANT != 1 || TAP) && tile_is_sky(p, tx, ty, tflag)) {
        if (VARIANT == 1) {
            const uint2 gsky = ld_guide(p.guide, x, y);
            for (int sig = 0; sig < NSIG; sig++)
                st_stream<uint4>(p.tapB[(HAS_SPEC && sig == SIG_SPEC) ? 1 : 0], x, y, 16, uint4{gsky.x, gsky.y, 0u, 0u});
        }
        return;
    }
    uint4 ctap[NSIG];
    if (TAP) {
#pragma unroll
        for (int sig = 0; sig < NSIG; sig++) {
            // one 16-byte buffer load per texel (NRD_CENTRE_B128): left to the compiler the load is split - the guide half in front of the per-pixel
            // sky test, the signal half behind it with 64-bit address arithmetic - and the set-up waits for a second round trip
#ifndef NRD_CENTRE_B128
#define NRD_CENTRE_B128 1
#endif
            if (NRD_CENTRE_B128)
                ctap[sig] = ldb<uint4>(plane_buf(tapIn[(HAS_SPEC && sig == SIG_SPEC) ? 1 : 0], 0), x, y, 16);
            else
                ctap[sig] = ld<uint4>(tapIn[(HAS_SPEC && sig == SIG_SPEC) ? 1 : 0], x, y, 16);
        }
    }
    const uint2 gtex = TAP ? uint2{ctap[0].x, ctap[0].y} : ld_guide(p.guide, x, y);
    Guide g = decode_guide(gtex, c.denoisingRange);
    if (g.sky) {
        for (int sig = 0; sig < NSIG; sig++) {
            if (TAP && VARIANT == 1) { // the guide part travels on (PostBlur takes its sky test from it)
                st_stream<uint4>(p.tapB[(HAS_SPEC && sig == SIG_SPEC) ? 1 : 0], x, y, 16, uint4{ctap[sig].x, ctap[sig].y, 0u, 0u});
                continue;
            }
            if (FUSED) // nobody reads Tmp1 in the fused frame
                continue;
            st<uint2>(outP, x, y, RBPT, uint2{0u, 0u}, sig * sb);
            if (SH)
                st<uint2>(outP, x, y, RBPT, uint2{0u, 0u}, sig * sb + 8);
        }
        if (VARIANT == 0 && HAS_SPEC)
            st<uint16_t>(p.hitTrack, x, y, 2, (uint16_t)0);
        if constexpr (FUSED)
            ta_sky_stores<HAS_DIFF, HAS_SPEC, false, false>(p, x, y);
        return;
    }
    // the specular signal's roughness terms: requested now, needed behind the diffuse signal's set-up
    uint4 rtRaw = {0u, 0u, 0u, 0u};
    if (NRD_ROUGH_LUT && HAS_SPEC)
        rtRaw = rough_terms_load(p.roughLut, gtex.x);
    const int gy0 = y + c.yOff;
    PixelGeo pg = pixel_geo(c, g, x, gy0, p.planeDistanceSensitivity);
    const NormalCodes ncodes = normal_codes(g.nw); // the taps' normal weights work on the 10-bit codes (nrd_device.h normal_cos)
    f3 V = to_viewer(pg.Xv);
    // Poisson rotation: per frame for PrePass / PostBlur - the 64 lanes of a wave (16x4 pixels) then gather 16x4-shaped texel
    // groups that coalesce into a few cache lines instead of 64 L1 lookups per load; per 2x2 quad for Blur (decorrelation; the lanes of a quad share cache lines)
    constexpr bool PER_PIXEL = VARIANT == 1;
    uint32_t h = hash_px(PER_PIXEL ? (uint32_t)x >> BLUR_ROTATION_SHIFT : 0u, PER_PIXEL ? (uint32_t)gy0 >> BLUR_ROTATION_SHIFT : 0u, c.frameIndex, (uint32_t)VARIANT + 1u); // one rotation per 2x2 quad
    float rc = c.rot[h & 63u][0], rs = c.rot[h & 63u][1];
    float diffA = 0.0f, specA = 0.0f;
    if (VARIANT != 0)
        unpack_data1(ld_stream<uint16_t>(p.data1, x, y, 2), diffA, specA);
    // tap window (global pixel coordinates): within `reach` of the centre, inside the frame and inside the held rows
    const int loX = imax(x - reach, 0), hiX = imin(x + reach, c.W - 1);
    const int loY = imax(gy0 - reach, imax(c.yOff, 0)), hiY = imin(gy0 + reach, imin(c.yOff + c.resH, c.H) - 1);
    const float loXf = (float)loX, hiXf = (float)hiX, loYf = (float)loY, hiYf = (float)hiY; // floored tap positions are compared / clamped as floats

    // ---- per-signal set-up: everything the taps of a signal need, for BOTH signals, before the first gather is issued -------
    // (a signal whose radius is 0 keeps its constants - its taps land on the centre and are selected out by `active`, which
    // leaves sum = centre, wsum = 1 exactly as if they had never run)
    const PlaneRef* srcPs[NSIG];
    const PlaneRef* src1Ps[NSIG];
    int srcOffs[NSIG];
    uint32_t matFloor[NSIG], matClass[NSIG]; // material test of the taps (nrd_device.h material_class)
    f4 sum[NSIG], sum1[NSIG];
    float wsum[NSIG], minHit[NSIG], hitNormS[NSIG];
    float jtx[NSIG], jty[NSIG], jbx[NSIG], jby[NSIG], m2w2[NSIG], hitA[NSIG], hitB[NSIG], roughA[NSIG], roughB[NSIG];
    bool active[NSIG];
    float kuRoughA = 0.0f; // (the specular signal's KernelUnit::roughA as computed: the fused kernel's TemporalAccumulation half uses it again)
    constexpr int srcBpt = VARIANT == 0 ? 8 : RBPT;
#pragma unroll
    for (int sig = 0; sig < NSIG; sig++) {
        const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
        float rough = isSpec ? g.roughness : 1.0f;
        matFloor[sig] = material_floor(isSpec ? p.minMatSpec : p.minMatDiff);
        matClass[sig] = material_class(g.mat, matFloor[sig]);
        srcPs[sig] = VARIANT == 0 ? (isSpec ? &p.inSpec : &p.inDiff) : &inP;
        srcOffs[sig] = VARIANT == 0 ? 0 : sig * sb;
        f4 center = TAP ? unpack_h4(uint2{ctap[sig].z, ctap[sig].w}) : load_signal(p, *srcPs[sig], x, y, srcBpt, srcOffs[sig], occIn);
        if (relaxIn && !RELAX_LINEAR_RGB)
            center = rgb_to_ycocg4(center);
        // SH mode: the SH1 texel rides along with exactly the weights of SH0 (separate IN_*_SH1 plane in the PrePass)
        src1Ps[sig] = VARIANT == 0 ? (isSpec ? &p.inSpec1 : &p.inDiff1) : &inP;
        sum1[sig] = SH ? unpack_h4(ld<uint2>(*src1Ps[sig], x, y, srcBpt, srcOffs[sig] + (VARIANT == 0 ? 0 : 8))) : f4{0, 0, 0, 0};
        // the pass-independent part of the set-up (taps are placed on the tangent plane linearised at the centre: KernelUnit::j)
        if (NRD_ROUGH_LUT && isSpec && HAS_DIFF)
            __builtin_amdgcn_sched_barrier(0); // (the table entry is consumed here, behind everything that does not need it)
        const RoughTerms rt = !isSpec ? RoughTerms{0.0f, 0.0f, 0.0f, 0.0f} : (NRD_ROUGH_LUT ? rough_terms_of(rtRaw) : rough_terms(p.hp, rough));
        const KernelUnit ku = isSpec ? kernel_unit<true>(p, pg, g.z, V, rough, rt) : kernel_unit<false>(p, pg, g.z, V, rough, rt);
        float hitNorm = fma_(pg.absZ, p.hp[1], p.hp[0]) * ku.hitFactor;
        float hitDist = center.w * hitNorm;
        float hitDistFactor = sat(hitDist * rcp_(pg.frustumSize));
        float A = isSpec ? specA : diffA;
        float nonLin = VARIANT == 0 ? 1.0f : rcp_(1.0f + A);
        float smc = ku.smc;
        float radius;
        if (VARIANT == 0) {
            radius = (isSpec ? p.specularPrepassBlurRadius : p.diffusePrepassBlurRadius) * hitDistFactor * smc;
        } else {
            float r = fma_(p.maxBlurRadius * lerpf(MIN_CONVERGED_RADIUS_SCALE, 1.0f, nonLin), lerpf(hitDistFactor, 1.0f, nonLin), p.minBlurRadius);
            r *= VARIANT == 2 ? POST_BLUR_RADIUS_SCALE : 1.0f;
            r *= smc;
            radius = p.maxBlurRadius != 0.0f ? r : 0.0f;
        }
        sum[sig] = center;
        wsum[sig] = 1.0f;
        minHit[sig] = center.w; // (PrePass: hit distance tracking, in units of the signal's w channel until the end)
        hitNormS[sig] = hitNorm;
        active[sig] = radius > 0.0f;
        jtx[sig] = ku.j[0] * radius, jty[sig] = ku.j[1] * radius;
        jbx[sig] = ku.j[2] * radius, jby[sig] = ku.j[3] * radius;
        if (PER_PIXEL) { // per-pixel rotation folded into the Jacobian (J . R): the taps then are the unrotated disk, 4 fma per tap
            const float a = fma_(rc, jtx[sig], rs * jbx[sig]), b = fma_(rc, jbx[sig], -(rs * jtx[sig]));
            const float cc = fma_(rc, jty[sig], rs * jby[sig]), d = fma_(rc, jby[sig], -(rs * jty[sig]));
            jtx[sig] = a;
            jbx[sig] = b;
            jty[sig] = cc;
            jby[sig] = d;
        }
        float angle = ku.angle0 * lerpf(p.lobeAngleFraction, 1.0f, nonLin);
        // (weight-class reciprocals from here on - wrcp_: the hwt flavour's v_rcp_f32; everything above feeds the tap coordinates and stays exact)
        float normalW = wrcp_(fmax2(angle, NORMAL_ANGLE_MIN));
        normalW *= strand_normal_relax(c, g.mat, absf(g.z)); // CommonSettings::strandMaterialID: thin strands relax the normal test
        m2w2[sig] = nw_param_m2(normalW);
        float hitScale = relaxIn ? wrcp_(fmax2(center.w, 1e-3f)) : 1.0f; // RELAX hit distances are world units: compare relatively
        hitA[sig] = hitScale * wrcp_(lerpf(1e-6f, 1.0f, fmin2(nonLin, smc))) * EXP_WEIGHT_SCALE; // (the exponent's scale folded in: exp_weight_prescaled)
        hitB[sig] = -center.w * hitA[sig];
        roughA[sig] = ku.roughA;
        if (isSpec)
            kuRoughA = ku.roughA;
        roughB[sig] = -rough * roughA[sig];
        if (TAP)
            roughA[sig] = roughA[sig] * (1.0f / 1023.0f); // applies to the tap's roughness CODE
    }

    // ---- tap loop: ONE software pipeline over the 8 taps of every signal ----------------------------------------------------
    // A tap has two halves: ISSUE (position on the linearised tangent plane, window test, the gathers of its guide and radiance
    // texels - clamped, always valid addresses) and CONSUME (weights + accumulation). Measured on MI355X (diagnosis
    // builds): the gathers of a pass cost ~0.17 ms of texture-addresser time and the arithmetic ~0.20 ms of VALU time, and with
    // "all gathers of a signal, then all arithmetic" the waves of a CU march in step - everybody gathers, then everybody computes -
    // so the two added up (0.30 ms). Here every wave keeps NRD_PIPE_DEPTH taps in flight and consumes tap T right after issuing
    // tap T + DEPTH, across the signal boundary, so each wave feeds the addresser and the VALU all the time.
    // Guide texels are NOT staged in LDS: with the XCD column-band tile traversal (nrd_device.h, xcd_tile) the gathers hit in L1 /
    // L2 and a staged 32 x 32 guide window measures the same 0.217 ms as global gathers (profiles/r02_ab_tile_traversal.txt; it was
    // worth 0.05 ms under the old round-robin traversal, profiles/r02_ab_blur_lds_postblur_waveshape.txt).
    // The tap window [lo, hi] folds the frame bounds, the rows this instance holds and the hard reach of the pass into one range
    // test per axis; a rejected tap is SELECTED out (sums untouched), exactly like an early "continue".
    // tap rows are GLOBAL rows: the band's first row is folded into the base pointers once (scalar unit) instead of one
    // subtract per tap (denoise_parts checks that (first row + rows held) x pitch stays a 32-bit offset)
    const bool anyRadius = VARIANT == 0 ? (p.diffusePrepassBlurRadius > 0.0f || p.specularPrepassBlurRadius > 0.0f) : p.maxBlurRadius != 0.0f;
    if (anyRadius) { // uniform (kernel argument)
        constexpr int NT = 8 * NSIG;
        // PrePass (hit-distance tracking state) and the SH flavours (a second radiance texel per tap) carry more registers per tap
        // Re-tuned at the end of round 3 (profiles/r03_ab_pipeline_depth.txt): with the 8-byte guide and the tap texels the radiance PrePass
        // is fastest with 2 taps in flight (78 VGPRs, 6 waves per SIMD: -6 % against 5 taps / 4 waves) and PostBlur with 4 (-2.3 %); Blur
        // stays at 8; RELAX's radiance PrePass: 3 (-3 %). The SH and OCCLUSION flavours keep round 2's depths (SH Blur at 3 or 2: +6 %).
        constexpr int DEPTH_WANTED = FUSED ? NRD_FUSED_DEPTH : (VARIANT == 0 && MODE == 0) ? NRD_PRE_DEPTH : (VARIANT == 0 && MODE == 1) ? 3 : (VARIANT == 0 && MODE == 4) ? NRD_RELAX_SH_PRE_DEPTH : (VARIANT == 0 && SH) ? NRD_SH_PRE_DEPTH : ((VARIANT == 0 || SH) ? NRD_PIPE_DEPTH_WIDE : (TAP ? (VARIANT == 2 ? NRD_POST_DEPTH : NRD_TAP_DEPTH) : NRD_PIPE_DEPTH));
        constexpr int DEPTH = DEPTH_WANTED < NT ? DEPTH_WANTED : NT;
        const PlaneBuf guideB = plane_buf(p.guide, c.yOff);
        PlaneBuf srcB[NSIG], src1B[NSIG];
#pragma unroll
        for (int sig = 0; sig < NSIG; sig++) {
            srcB[sig] = plane_buf(TAP ? tapIn[(HAS_SPEC && sig == SIG_SPEC) ? 1 : 0] : *srcPs[sig], c.yOff);
            if (SH)
                src1B[sig] = plane_buf(*src1Ps[sig], c.yOff);
        }
        const float cx = (float)x + 0.5f, cy = (float)gy0 + 0.5f;
        float gaT[NT]; // plane-equation term of geo_weight at the tap position
        bool inWin[NT];
        uint4 graw[NT];
        uint2 sraw[NT], sraw1[NT];
        // Tap order: signal-major (the 8 taps of a signal, then the next signal's)
        auto sig_of = [&](const int T) { return T >> 3; };
        auto tap_of = [&](const int T) { return T & 7; };
        auto tap_offset = [&](const int t, float& ox, float& oy) {
            if (PER_PIXEL) {
                ox = g_poisson8[t][0];
                oy = g_poisson8[t][1];
            } else {
                ox = VARIANT == 0 ? p.tapsPre[t][0] : p.tapsPost[t][0];
                oy = VARIANT == 0 ? p.tapsPre[t][1] : p.tapsPost[t][1];
            }
        };
        auto gather = [&](const int T, const float fpx, const float fpy) {
            const int sig = sig_of(T);
            // inside the (never empty) window <=> clamping leaves the position unchanged; NaN positions compare unequal
            const float cxf = __builtin_amdgcn_fmed3f(fpx, loXf, hiXf), cyf = __builtin_amdgcn_fmed3f(fpy, loYf, hiYf);
            inWin[T] = (cxf == fpx) & (cyf == fpy);
            const int px = (int)cxf, gpy = (int)cyf;
            if constexpr (TAP) { // ONE gather: {guide part | signal}
                graw[T] = ldb<uint4>(srcB[sig], px, gpy, 16);
                return;
            }
            {
                const uint2 gt = ldb<uint2>(guideB, px, gpy, GUIDE_BYTES);
                graw[T] = uint4{gt.x, gt.y, 0u, 0u};
            }
            if constexpr (SH && VARIANT != 0) { // SH0 | SH1 of a signal sit side by side in the internal planes: ONE 16-byte gather
                const uint4 both = ldb<uint4>(srcB[sig], px, gpy, srcBpt, srcOffs[sig]);
                sraw[T] = uint2{both.x, both.y};
                sraw1[T] = uint2{both.z, both.w};
            } else {
                sraw[T] = occIn ? uint2{(uint32_t)ldb<uint16_t>(srcB[sig], px, gpy, 2), 0u} : ldb<uint2>(srcB[sig], px, gpy, srcBpt, srcOffs[sig]);
                sraw1[T] = SH ? ldb<uint2>(src1B[sig], px, gpy, srcBpt, srcOffs[sig]) : uint2{0u, 0u};
            }
        };
        auto issue = [&](const int T) {
            const int sig = sig_of(T);
            float ox, oy;
            tap_offset(tap_of(T), ox, oy);
            const float fpx = __builtin_floorf(fma_(ox, jtx[sig], fma_(oy, jbx[sig], cx)));
            const float fpy = __builtin_floorf(fma_(ox, jty[sig], fma_(oy, jby[sig], cy)));
            gaT[T] = fma_(pg.gax, fpx, fma_(pg.gay, fpy, pg.ga0));
            gather(T, fpx, fpy);
        };
        // a tap's texels -> its guide fields and its signal
        auto decode = [&](const int T, Guide& gs, f4& sv) {
            if constexpr (TAP) {
                gs.z = u2f(graw[T].x);
                gs.mat = graw[T].y >> 30;
                gs.nw = graw[T].y;
                gs.sky = !(absf(gs.z) <= c.denoisingRange);
                sv = unpack_h4(uint2{graw[T].z, graw[T].w});
            } else {
                gs = decode_guide(uint2{graw[T].x, graw[T].y}, c.denoisingRange);
                sv = decode_signal(p, sraw[T], occIn);
            }
            if (relaxIn && !RELAX_LINEAR_RGB)
                sv = rgb_to_ycocg4(sv);
        };
        auto tap_valid = [&](const int T, const Guide& gs) {
            const int sig = sig_of(T);
            const bool matOk = material_class(gs.mat, matFloor[sig]) == matClass[sig];
            return (bool)(inWin[T] & active[sig] & !gs.sky & matOk); // bitwise: one basic block
        };
        // roughness weight of a specular tap (the tap texels carry the roughness CODE in the low bits of the depth word)
        auto rough_weight = [&](const int T, const Guide& gs) {
            const float r = TAP ? (float)(graw[T].x & 1023u) : gs.roughness;
            constexpr int SS = HAS_SPEC ? NSIG - 1 : 0; // the specular signal is the last one
            return smoothstep01(1.0f - absf(fma_(r, roughA[SS], roughB[SS])));
        };
        auto accumulate = [&](const int T, const f4 sv, float w, const bool valid) {
            const int sig = sig_of(T);
            if (VARIANT == 0) {
                // PrePass reads caller-owned inputs (garbage allowed on sky / outside the rect): a rejected tap is
                // selected out component by component (a straight-line form - the texel of a rejected tap zeroed before it is decoded, one
                // select on the weight - measured +0.2 % on the headline and -2 % on RELAX SH's PrePass: tools/variants/window_fast_path_prepass_straight.patch)
                f4 acc = fma4(sv, w, sum[sig]);
                sum[sig] = {valid ? acc.x : sum[sig].x, valid ? acc.y : sum[sig].y, valid ? acc.z : sum[sig].z, valid ? acc.w : sum[sig].w};
                if (SH) {
                    f4 acc1 = fma4(unpack_h4(sraw1[T]), w, sum1[sig]);
                    sum1[sig] = {valid ? acc1.x : sum1[sig].x, valid ? acc1.y : sum1[sig].y, valid ? acc1.z : sum1[sig].z, valid ? acc1.w : sum1[sig].w};
                }
                wsum[sig] = valid ? wsum[sig] + w : wsum[sig];
                // tracked hit distance: the smallest hit distance CODE among the taps that count (the scale is applied once, at the end: a
                // product with a positive constant is monotone under rounding, so the minimum commutes with it exactly)
                minHit[sig] = (valid & (w > 0.0f)) ? fmin2(minHit[sig], sv.w) : minHit[sig];
            } else {
                // Blur / PostBlur read internal planes (always finite): a rejected tap enters with weight 0 - one select
                w = valid ? w : 0.0f;
                sum[sig] = fma4(sv, w, sum[sig]);
                if (SH)
                    sum1[sig] = fma4(unpack_h4(sraw1[T]), w, sum1[sig]);
                wsum[sig] += w;
            }
        };
        auto consume = [&](const int T) {
            const int sig = sig_of(T), t = tap_of(T);
            const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
            Guide gs;
            f4 sv;
            decode(T, gs, sv);
            const bool valid = tap_valid(T, gs);
            float w = g_poisson8[t][2];
            w *= smoothstep01(1.0f - absf(geo_plane(pg, gaT[T], gs.z))); // == geo_weight(pg, fpx, fpy, gs.z)
            w *= normal_weight_m2(normal_dist2(ncodes, gs.nw), m2w2[sig]);
            if (isSpec)
                w *= rough_weight(T, gs);
            w *= lerpf(p.minHitDistanceWeight, 1.0f, exp_weight_prescaled(fma_(sv.w, hitA[sig], hitB[sig])));
            accumulate(T, sv, w, valid);
        };
        // (round 6: a second copy of the loop without the window test for waves whose taps are provably inside - 4 of ~66 instructions per
        // tap - measured +0.4 % at best and costs Blur 9 registers: profiles/r06_ab_window_fast_path.txt, tools/variants/window_fast_path_prepass_straight.patch)
        {
#pragma unroll
            for (int T = 0; T < DEPTH; T++)
                issue(T);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int T = 0; T < NT; T++) {
                if (T + DEPTH < NT)
                    issue(T + DEPTH);
                __builtin_amdgcn_sched_barrier(0);
                consume(T);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    uint2 outw[RBPT / 8];
#pragma unroll
    for (int sig = 0; sig < NSIG; sig++) {
        const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
        float invw = wrcp_(wsum[sig]);
        f4 res = mul4(sum[sig], invw), res1 = mul4(sum1[sig], invw);
        if (VARIANT == 0 && isSpec && p.prepassTrackOnly) { // pass-through: the centre is fetched again instead of being kept live across the tap loop
            res = load_signal(p, *srcPs[sig], x, y, srcBpt, srcOffs[sig], occIn);
            if (relaxIn && !RELAX_LINEAR_RGB)
                res = rgb_to_ycocg4(res);
            res1 = SH ? unpack_h4(ld<uint2>(*src1Ps[sig], x, y, srcBpt, srcOffs[sig] + (VARIANT == 0 ? 0 : 8))) : f4{0, 0, 0, 0};
        }
        // Blur / PostBlur results are not read again before the gathers of this launch are done (PostBlur's only by the next frame):
        // streamed past the caches, they stop evicting the tap texels (PostBlur -8 %, Blur -4 %). The PrePass result is read by the
        // next kernel at once and travels through the Infinity Cache: a plain store (profiles/r03_ab_setup_planes.txt)
        if (TAP && VARIANT == 1) {
            const uint2 rv = pack_h4(res);
            st_stream<uint4>(p.tapB[isSpec ? 1 : 0], x, y, 16, uint4{ctap[sig].x, ctap[sig].y, rv.x, rv.y});
        } else {
            outw[sig * (sb / 8)] = pack_h4(res);
            if (SH)
                outw[sig * (sb / 8) + 1] = pack_h4(res1);
        }
        if (VARIANT == 0 && isSpec) {
            minHit[sig] = minHit[sig] * hitNormS[sig];
            st<uint16_t>(p.hitTrack, x, y, 2, f2h(minHit[sig])); // (fused frame: TemporalStabilization still reads it)
        }
    }
    if constexpr (FUSED) { // on into TemporalAccumulation with what Tmp1 and the hit tracker would have held
#ifndef NRD_FUSED_RELOAD_ARGS
#define NRD_FUSED_RELOAD_ARGS 1
#endif
        if (NRD_FUSED_RELOAD_ARGS) {
            NRD_RELOAD_ARGS(ReblurParams, p, q); // (nrd_device.h: the reprojection's constants are loaded behind the tap loop, not held across it)
            const TaGeo tg = {pg.Xv, pg.Nv, V, (HAS_SPEC && !HW_TRANSCENDENTALS) ? kuRoughA : -1.0f};
            ta_pixel<HAS_DIFF, HAS_SPEC, false, false, NRD_FUSED_SEQ_FOOTPRINTS != 0>(q, x, y, g, outw, HAS_SPEC ? h2f(f2h(minHit[SIG_SPEC])) : 0.0f, NRD_TA_LEAN >= 3 ? nullptr : &tg);
        } else
            ta_pixel<HAS_DIFF, HAS_SPEC, false, false, NRD_FUSED_SEQ_FOOTPRINTS != 0>(p, x, y, g, outw, HAS_SPEC ? h2f(f2h(minHit[SIG_SPEC])) : 0.0f);
        return;
    }
    // the signals of the pixel share one texel of the output plane: one store
    if (VARIANT == 0)
        store_texel<RBPT, false>(outP, x, y, outw);
    else if (!(TAP && VARIANT == 1))
        store_texel<RBPT, true>(outP, x, y, outw);
}

template <int VARIANT, int MODE, bool HAS_DIFF, bool HAS_SPEC>
// 4 waves per SIMD (<= 128 VGPRs, a handful of spilled dwords) beat 3 waves with everything in registers; the SH flavours
// carry 16 more registers of tap data and stay at 3
__global__ __launch_bounds__(256) NRD_WAVES_PER_EU(MODE >= 3 ? (VARIANT == 0 ? (MODE == 4 ? NRD_RELAX_SH_PRE_WAVES : NRD_SH_PRE_WAVES) : 3) : ((VARIANT != 0 && MODE == 0) ? (VARIANT == 2 ? NRD_POST_WAVES : NRD_TAP_WAVES) : (VARIANT == 0 ? NRD_PRE_WAVES : 4))) void k_spatial(const ReblurParams p) {
    if (NRD_PIN_ARGS) { // the planes of the first vector loads (nrd_device.h NRD_PIN_SGPRS8)
        if (VARIANT != 0 && MODE == 0) {
            const PlaneRef* tapIn = VARIANT == 1 ? p.tapA : p.tapB;
            NRD_PIN_PLANES3(tapIn[0], tapIn[HAS_DIFF && HAS_SPEC ? 1 : 0], p.data1);
        } else
            NRD_PIN_PLANES3(p.guide, HAS_DIFF ? p.inDiff : p.inSpec, p.inSpec);
    }
    int x, y, tx, ty, tflag;
    if (!my_pixel(p.c, x, y, tx, ty, tflag)) // (one wave per workgroup measured 6-16 % SLOWER here: the four quarters of a tile land on
        return;                              // four CUs and stop sharing an L1 - profiles/r02_ab_tile_traversal.txt)
    // (round 4: workgroups that take 2 / 4 consecutive tiles with the next tile's centre loads in flight behind the current tile's
    // arithmetic measured 14 / 24 % SLOWER on Blur and 10 / 18 % on PostBlur - profiles/r04_ab_fusion.txt)
    spatial_pixel<VARIANT, MODE, HAS_DIFF, HAS_SPEC, false>(p, x, y, tx, ty, tflag);
}

// =====================================================================================================================
// Reprojection helpers (TemporalAccumulation + TemporalStabilization)
// =====================================================================================================================
struct Reproj {
    float su, sv;
    f3 Xw, XwPrev, XvPrev;
    float zPrev;
};

NRD_DEV Reproj reproject(const FrameConsts& c, f3 Xv, float u, float v, f4 mvRaw) {
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
            r.XwPrev = add3(rot3(c.v2wPrev, r.XvPrev), cd);
        } else {
            r.XwPrev = r.Xw;
            r.XvPrev = rot3(c.w2vPrev, sub3(r.XwPrev, cd));
            r.zPrev = r.XvPrev.z;
        }
    }
    return r;
}

struct Footprint {
    int ix, iy;
    float w[4];
    float wsum;
    uint32_t bits;
};

// `mat` = material of the reflecting pixel: reflections seen in surfaces of CommonSettings::cameraAttachedReflectionMaterialID
// (Source/NRDSample.cpp:3869-3876: objects that travel with the camera, e.g. a first-person weapon) keep their VIEW-space
// position from frame to frame, so their virtual point is projected as it stands in the current view instead of being carried
// back through the camera motion. The test is wave-uniform off when no material is named (the sample's setting, Shared.hlsli:44).
// (`f` = spec_dominant_factor of the pixel's roughness: dominant_factor_of below)
NRD_DEV bool virtual_uv(const FrameConsts& c, const Reproj& r, float hitDist, float f, uint32_t mat, float& vu, float& vv) {
    f3 toCam = ORTHO ? rot3(c.v2w, f3{0.0f, 0.0f, r.zPrev >= 0.0f ? 1.0f : -1.0f}) : normalize3(r.Xw); // direction camera -> surface
    f3 Xvirt = add3(r.Xw, mul3(toCam, hitDist * f));
    f3 XvirtPrev = add3(Xvirt, sub3(r.XwPrev, r.Xw));
    f3 rel = sub3(XvirtPrev, {c.camDelta[0], c.camDelta[1], c.camDelta[2]});
    f3 Xp = rot3(c.w2vPrev, rel);
    if (c.camAttachMat != 0xffffffffu) {
        f3 Xa = rot3(c.w2v, Xvirt);
        bool attached = mat == c.camAttachMat;
        Xp = {attached ? Xa.x : Xp.x, attached ? Xa.y : Xp.y, attached ? Xa.z : Xp.z};
    }
    return project(c.pjPrev, Xp, vu, vv);
}

NRD_DEV float sample_confidence(const PlaneRef& P, float u, float v) {
    if (!P.p)
        return 1.0f;
    float px = fma_(u, (float)P.w, -0.5f), py = fma_(v, (float)P.h, -0.5f);
    float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
    float fx = px - fx0, fy = py - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    int xa = x0 < 0 ? 0 : (x0 >= P.w ? P.w - 1 : x0), xb = x0 + 1 < 0 ? 0 : (x0 + 1 >= P.w ? P.w - 1 : x0 + 1);
    int ya = y0 < 0 ? 0 : (y0 >= P.h ? P.h - 1 : y0), yb = y0 + 1 < 0 ? 0 : (y0 + 1 >= P.h ? P.h - 1 : y0 + 1);
    float a = lerpf(h2f(ld<uint16_t>(P, xa, ya, 8)), h2f(ld<uint16_t>(P, xb, ya, 8)), fx);
    float b = lerpf(h2f(ld<uint16_t>(P, xa, yb, 8)), h2f(ld<uint16_t>(P, xb, yb, 8)), fx);
    return sat(lerpf(a, b, fy));
}

// (`at` = the roughness-only terms {1 - 2^(-200 r^2), log2(r)} from the table - accum_terms_of below; nullptr: evaluate them)
NRD_DEV float spec_accum_limit(float roughness, float NoV, float parallaxPx, const float2* at = nullptr) {
    float acos01sq = sat(1.0f - NoV * 0.99999f);
    float a = sqrt_(acos01sq);
    float b = fma_(roughness, roughness, 1.1f);
    float parallaxSensitivity = (b + a) * rcp_(b - a);
    float powerScale = fma_(parallaxSensitivity * parallaxPx, 2.0f, 1.0f);
    if (at) { // pow01(r, y) = r <= 0 ? 0 : 2^(y log2 r), r in [0, 1] already
        const float pw = roughness <= 0.0f ? 0.0f : exp2_poly((0.5f * powerScale) * at->y);
        return MAX_ACCUM * (at->x * pw);
    }
    float f = 1.0f - exp2_poly(-200.0f * roughness * roughness);
    f *= pow01(roughness, 0.5f * powerScale);
    return MAX_ACCUM * f;
}

// =====================================================================================================================
// K3 TemporalAccumulation
// =====================================================================================================================
// Everything TemporalAccumulation needs from ONE bilinear footprint of the previous frame, fetched in one go: the four texel
// positions depend only on the reprojected uv, so guides, accumulation speeds, radiance history and luma histories of all
// signals are gathered unconditionally (clamped addresses) BEFORE any of them is validated - the surface-motion and the
// virtual-motion footprints together are one memory round trip instead of a chain of dependent ones.
template <int RBPT, int LBPT, bool RELAX>
struct FootRaw {
    uint2 g[4];
    uint16_t a[4];
    uint2 t[4][RBPT / 8];
    uint32_t f[4]; // fast luma history texel: LBPT bytes = one fp16 per signal
    uint32_t m[4]; // RELAX: second luma moment history
};
NRD_DEV uint32_t load_luma(const PlaneRef& P, int x, int y, int lbpt) { return lbpt == 4 ? ld<uint32_t>(P, x, y, 4) : (uint32_t)ld<uint16_t>(P, x, y, 2); }
NRD_DEV void store_luma(const PlaneRef& P, int x, int y, int lbpt, uint32_t v) {
    if (lbpt == 4)
        st<uint32_t>(P, x, y, 4, v);
    else
        st<uint16_t>(P, x, y, 2, (uint16_t)v);
}

struct FootPos {
    int ix, iy;
    float fx, fy;
    bool sane;
};
NRD_DEV FootPos foot_pos(const FrameConsts& c, float pu, float pv) {
    FootPos f;
    float px = fma_(pu, (float)c.Wprev, -0.5f), py = fma_(pv, (float)c.Hprev, -0.5f);
    float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
    f.fx = px - fx0;
    f.fy = py - fy0;
    f.sane = fx0 >= -2.0f && fx0 <= (float)c.Wprev + 1.0f && fy0 >= -2.0f && fy0 <= (float)c.Hprev + 1.0f;
    f.ix = f.sane ? (int)fx0 : -4;
    f.iy = f.sane ? (int)fy0 : -4;
    return f;
}
template <int RBPT, int LBPT, bool RELAX>
NRD_DEV void load_foot(const ReblurParams& p, const FootPos& fp, FootRaw<RBPT, LBPT, RELAX>& r) {
    // one load per texel and plane. Fetching the two texels of a footprint row with ONE texel-aligned wide load (16 bytes at an 8-byte
    // boundary, a dword at a 2-byte one: 20 memory instructions instead of 32) works and is bit-identical, but the addresser splits
    // such accesses: TemporalAccumulation +10 %, TemporalStabilization +12 % (profiles/r03_ab_staging_and_pair_loads.txt)
    const FrameConsts& c = p.c;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int tx = imin(imax(fp.ix + (i & 1), 0), c.Wprev - 1), ty = imin(imax(fp.iy + (i >> 1) - c.yOff, 0), c.resH - 1);
        r.g[i] = ld_guide(p.guidePrev, tx, ty);
        r.a[i] = ld<uint16_t>(p.data1Prev, tx, ty, 2);
        load_texel<RBPT>(p.hist, tx, ty, r.t[i]);
        r.f[i] = load_luma(p.fastPrev, tx, ty, LBPT);
        r.m[i] = RELAX ? load_luma(p.stabPrev, tx, ty, LBPT) : 0u;
    }
}
// validation of the four texels of a footprint
NRD_DEV Footprint foot_weights(const FrameConsts& c, const FootPos& fp, const uint2 (&graw)[4], f3 NvPrev, f3 XvPrev, f3 N, uint32_t mat, uint32_t minMat, float threshold) {
    Footprint f;
    f.ix = fp.ix;
    f.iy = fp.iy;
    float fx = fp.fx, fy = fp.fy;
    float bw[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
    f.wsum = 0.0f;
    f.bits = 0;
    float planeRef = dot3(NvPrev, XvPrev);
    float g0 = ORTHO ? fma_(NvPrev.x, c.pvPrev[0], NvPrev.y * c.pvPrev[1]) : fma_(NvPrev.x, c.pvPrev[0], fma_(NvPrev.y, c.pvPrev[1], NvPrev.z));
    float gx = NvPrev.x * c.pvPrev[2], gyc = NvPrev.y * c.pvPrev[3];
    const uint32_t mfloor = material_floor(minMat), mclass = material_class(mat, mfloor);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int tx = f.ix + (i & 1), gy = f.iy + (i >> 1), ty = gy - c.yOff;
        bool ok = fp.sane & (tx >= 0) & (tx < c.Wprev) & (gy >= 0) & (gy < c.Hprev) & (ty >= c.prevY0) & (ty < c.prevY1);
        Guide gp = decode_guide(graw[i], c.denoisingRange);
        float lin = fma_(gx, (float)tx, fma_(gyc, (float)gy, g0));
        float plane = ORTHO ? fma_(gp.z, NvPrev.z, lin) : gp.z * lin; // N . X of the previous-frame texel
        // (bitwise: every operand is a plain comparison; the short-circuit form compiles to an exec-mask detour per operand)
        const bool planeOk = absf(plane - planeRef) <= threshold, normalOk = dot3(N, gp.n) > PREV_NORMAL_COS, matOk = material_class(gp.mat, mfloor) == mclass;
        ok = ok & !gp.sky & planeOk & normalOk & matOk;
        f.w[i] = ok ? bw[i] : 0.0f;
        f.wsum += f.w[i];
        f.bits |= ok ? (1u << i) : 0u;
    }
    return f;
}
// weighted texel blends of an already fetched footprint. A rejected texel has weight exactly 0 and every history plane holds
// finite values (clamped fp16 stores, zeros on sky / never-written texels), so accumulating it unconditionally adds +-0 to a sum
// that is never -0: bit-identical to skipping it (what the oracle does), without a select per component
template <int WORDS>
NRD_DEV f4 blend4(const Footprint& f, const uint2 (&t)[4][WORDS], int word) {
    f4 s = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++)
        s = fma4(unpack_h4(t[i][word]), f.w[i], s);
    return mul4(s, rcp_(f.wsum));
}
NRD_DEV float blend1(const Footprint& f, const uint32_t (&r)[4], int half) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++)
        s = fma_(h2f((uint16_t)(r[i] >> (16 * half))), f.w[i], s);
    return s * (rcp_(f.wsum));
}
NRD_DEV void blendA(const Footprint& f, const uint16_t (&raw)[4], float& dA, float& sA) {
    dA = sA = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float a, b;
        unpack_data1(raw[i], a, b);
        dA = fma_(a, f.w[i], dA);
        sA = fma_(b, f.w[i], sA);
    }
    float inv = rcp_(f.wsum);
    dA *= inv;
    sA *= inv;
}

#ifndef NRD_TA_WAVES // waves per SIMD the compiler budgets TemporalAccumulation's registers for (REBLUR radiance flavours use 107 VGPRs: 4)
#define NRD_TA_WAVES 4
#endif
// what TemporalAccumulation writes at a pixel beyond the denoising range (inside a tile that has geometry)
template <bool HAS_DIFF, bool HAS_SPEC, bool SH, bool RELAX>
NRD_DEV void ta_sky_stores(const ReblurParams& p, int x, int y) {
    constexpr int NSIG = (HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0);
    constexpr int sb = SH ? 16 : 8;
    constexpr int RBPT = sb * NSIG;
    constexpr int LBPT = 2 * NSIG;
    for (int sig = 0; sig < NSIG; sig++) {
        st<uint2>(p.tmp2, x, y, RBPT, uint2{0u, 0u}, sig * sb);
        if (SH)
            st<uint2>(p.tmp2, x, y, RBPT, uint2{0u, 0u}, sig * sb + 8);
        st<uint16_t>(p.fast, x, y, LBPT, (uint16_t)0, sig * 2);
        if (RELAX)
            st<uint16_t>(p.stab, x, y, LBPT, (uint16_t)0, sig * 2);
    }
    st<uint16_t>(p.data1Tmp, x, y, 2, (uint16_t)0);
    st<uint32_t>(p.data2, x, y, 4, 0u);
}

template <bool HAS_DIFF, bool HAS_SPEC, bool SH, bool RELAX, bool SEQ_FOOT>
NRD_DEV void ta_pixel(const ReblurParams& p, const int x, const int y, const Guide& g, const uint2 (&ctex)[((SH ? 16 : 8) * ((HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0))) / 8], const float hitDist, const TaGeo* geo) {
    constexpr int NSIG = (HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0);
    constexpr int sb = SH ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    constexpr int RBPT = sb * NSIG;
    constexpr int SW = sb / 8, S1 = SH ? 1 : 0; // uint2 words per signal; word of the SH1 texel
    constexpr int LBPT = 2 * NSIG;
    constexpr int SIG_SPEC = HAS_DIFF ? 1 : 0;
    const FrameConsts& c = p.c;
    f4 mvRaw = unpack_h4(ld<uint2>(p.inMV, x, y, 8));
    const int gy0 = y + c.yOff;
    float u = ((float)x + 0.5f) * c.invW, v = ((float)gy0 + 0.5f) * c.invH;
    float confD = (HAS_DIFF && c.confAvail) ? sample_confidence(p.confD, u, v) : 1.0f;
    // the sample binds ONE texture to both confidence slots (Source/NRDSample.cpp:457, :462): fetch it once then
    float confS = (HAS_SPEC && c.confAvail) ? ((HAS_DIFF && p.confS.p == p.confD.p) ? confD : sample_confidence(p.confS, u, v)) : 1.0f;
    f3 Xv = geo ? geo->Xv : reconstruct_px(c.pv, (float)x, (float)gy0, g.z);
    f3 Nv = geo ? geo->Nv : rot3(c.w2v, g.n);
    f3 V = geo ? geo->V : to_viewer(Xv);
    float NoV = absf(dot3(Nv, V));
    Reproj r = reproject(c, Xv, u, v, mvRaw);
    f3 NvPrev = rot3(c.w2vPrev, g.n);
    float thrBase = c.disocclusionThreshold;
    if (c.mixAvail) // per-pixel blend toward disocclusionThresholdAlternate
        thrBase = lerpf(thrBase, c.disoccAlt, (float)ld<uint8_t>(p.inMix, x, y, 1) * (1.0f / 255.0f));
    float threshold = thrBase * c.minRectDimMulUnproject * zpersp(absf(r.zPrev));
    uint32_t minMatAny = p.minMatDiff < p.minMatSpec ? p.minMatDiff : p.minMatSpec;
    const bool historyOk = c.historyOk != 0;
    const float domFactor = HAS_SPEC ? dominant_factor_of(p, g) : 0.0f;
    // ---- both footprints: positions, then ALL their gathers, then validation
    FootPos spos = foot_pos(c, r.su, r.sv);
    FootRaw<RBPT, LBPT, RELAX> sraw, vraw;
    float accumLimit = 0.0f;
    if (NRD_TA_LEAN && HAS_SPEC) {
        f3 cd = {c.camDelta[0], c.camDelta[1], c.camDelta[2]};
        f3 XparV = rot3(c.w2v, sub3(r.XwPrev, cd));
        float pu, pv, parallax = 0.0f;
        if (project(c.pj, XparV, pu, pv)) {
            float dx = (pu - r.su) * (float)c.W, dy = (pv - r.sv) * (float)c.H;
            parallax = sqrt_(fma_(dx, dx, dy * dy));
        }
        const float2 accumTerms = NRD_ROUGH_LUT ? accum_terms_of(p, g) : float2{0.0f, 0.0f};
        accumLimit = spec_accum_limit(g.roughness, NoV, parallax, NRD_ROUGH_LUT ? &accumTerms : nullptr);
    }
    if (!NRD_TA_LEAN)
        load_foot(p, spos, sraw);
    float vu = -10.0f, vv = -10.0f;
    bool vOk = false;
    if (HAS_SPEC) {
        float tu, tv;
        vOk = virtual_uv(c, r, hitDist, domFactor, g.mat, tu, tv) && historyOk;
        vu = vOk ? tu : -10.0f; // an unusable virtual position lands outside: no texel validates (bits 0, weight 0)
        vv = vOk ? tv : -10.0f;
    }
    FootPos vpos = foot_pos(c, vu, vv);
    if (NRD_TA_LEAN) {
        __builtin_amdgcn_sched_barrier(0);
        load_foot(p, spos, sraw);
    }
    // SEQ_FOOT (the fused PrePass + TemporalAccumulation kernel): the virtual-motion footprint is fetched AFTER everything that hangs on the
    // surface-motion one has been reduced to a handful of values - a second round trip per wave, but 32 registers of raw footprint texels
    // less: with 2 PrePass taps in flight the fused kernel fits 93 VGPRs = 5 waves per SIMD, 0.343 -> 0.326 ms (profiles/
    // r05_ab_sequential_footprints.txt). The stand-alone TemporalAccumulation kernels (SH / RELAX flavours, separate passes) wait for memory,
    // not for issue slots: there the second round trip costs 0-6 % and both footprints stay one trip (same file).
    constexpr bool SEQ = SEQ_FOOT && HAS_SPEC;
    if (HAS_SPEC && !SEQ)
        load_foot(p, vpos, vraw);
    Footprint smb = foot_weights(c, spos, sraw.g, NvPrev, r.XvPrev, g.n, g.mat, minMatAny, threshold);
    bool smbOk = historyOk && smb.wsum > 0.0f;
    float prevDiffA, prevSpecA;
    blendA(smb, sraw.a, prevDiffA, prevSpecA);
    prevDiffA = smbOk ? fmin2(prevDiffA + 1.0f, p.maxA) : 0.0f;
    prevSpecA = smbOk ? fmin2(prevSpecA + 1.0f, p.maxASpec) : 0.0f;
    float quality = smbOk ? smb.wsum : 0.0f;
    float outDiffA = 0.0f, outSpecA = 0.0f;
    uint32_t data2 = smbOk ? smb.bits : 0u;
    uint2 outw[RBPT / 8]; // the signals of the pixel share one texel of Tmp2 / of the fast history: they leave in one store each
    uint32_t fastw = 0u, m2w = 0u;

    if (HAS_DIFF) {
        f4 in = unpack_h4(ctex[0]);
        float A = prevDiffA;
        A *= confD;
        A *= lerpf(quality, 1.0f, rcp_(1.0f + A));
        float nonLin = rcp_(1.0f + A);
        f4 hist = smbOk ? blend4(smb, sraw.t, 0) : in;
        const float inY = signal_luma(in, RELAX);
        float fastHist = smbOk ? blend1(smb, sraw.f, 0) : inY;
        outw[0] = pack_h4(lerp4(hist, in, nonLin));
        if (SH) { // SH1 follows SH0: same footprint, same blend factor
            f4 in1 = unpack_h4(ctex[S1]);
            f4 hist1 = smbOk ? blend4(smb, sraw.t, S1) : in1;
            outw[S1] = pack_h4(lerp4(hist1, in1, nonLin));
        }
        fastw = (uint32_t)f2h(lerpf(fastHist, inY, rcp_(1.0f + fmin2(A, p.maxFastA))));
        if (RELAX) { // second luma moment history (lives in the stabilized-luma slots)
            float m2 = inY * inY;
            float m2prev = smbOk ? blend1(smb, sraw.m, 0) : m2;
            m2w = (uint32_t)f2h(lerpf(m2prev, m2, nonLin));
        }
        outDiffA = A;
    }
    if (HAS_SPEC) {
        constexpr int sw = SIG_SPEC * SW;
        constexpr int lo = SIG_SPEC * 2;
        f4 in = unpack_h4(ctex[sw]);
        if (!NRD_TA_LEAN) {
            f3 cd = {c.camDelta[0], c.camDelta[1], c.camDelta[2]};
            f3 XparV = rot3(c.w2v, sub3(r.XwPrev, cd));
            float pu, pv, parallax = 0.0f;
            if (project(c.pj, XparV, pu, pv)) {
                float dx = (pu - r.su) * (float)c.W, dy = (pv - r.sv) * (float)c.H;
                parallax = sqrt_(fma_(dx, dx, dy * dy));
            }
            const float2 accumTerms = NRD_ROUGH_LUT ? accum_terms_of(p, g) : float2{0.0f, 0.0f};
            accumLimit = spec_accum_limit(g.roughness, NoV, parallax, NRD_ROUGH_LUT ? &accumTerms : nullptr);
        }
        float Asmb = fmin2(prevSpecA, accumLimit);
        const float inY = signal_luma(in, RELAX);
        f4 smbHist = smbOk ? blend4(smb, sraw.t, sw) : in;
        float smbFast = smbOk ? blend1(smb, sraw.f, SIG_SPEC) : inY;
        f4 smb1 = {0, 0, 0, 0};
        if (SH)
            smb1 = smbOk ? blend4(smb, sraw.t, sw + S1) : unpack_h4(ctex[sw + S1]);
        float m2smb = 0.0f;
        if (RELAX)
            m2smb = smbOk ? blend1(smb, sraw.m, SIG_SPEC) : inY * inY;
        if (SEQ) {
            __builtin_amdgcn_sched_barrier(0);
            load_foot(p, vpos, vraw);
        }
        Footprint vmb = foot_weights(c, vpos, vraw.g, NvPrev, r.XvPrev, g.n, g.mat, p.minMatSpec, threshold);
        uint32_t vmbBits = vmb.bits;
        bool vmbOk = vmb.wsum > 0.0f;
        // roughness of the virtual footprint: guide texel bytes 10..11 = upper half of .z
        // roughness of the virtual footprint: the 10-bit code in the low bits of the guide texel's depth word
        float prevRough = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++)
            prevRough = fma_((float)(vraw.g[i].x & 1023u) * (1.0f / 1023.0f), vmb.w[i], prevRough);
        prevRough *= rcp_(vmb.wsum);
        float roughA = (geo && geo->roughA >= 0.0f) ? geo->roughA : rcp_(lerpf(0.01f, 1.0f, sat(g.roughness * p.roughnessFraction)));
        float rconf = smoothstep01(1.0f - absf((prevRough - g.roughness) * roughA));
        float amount = vmbOk ? domFactor * vmb.wsum * rconf : 0.0f;
        float dA, sA;
        blendA(vmb, vraw.a, dA, sA);
        float Avmb = vmbOk ? fmin2(sA + 1.0f, p.maxASpec) : 0.0f;
        f4 vmbHist = vmbOk ? blend4(vmb, vraw.t, sw) : in;
        float vmbFast = vmbOk ? blend1(vmb, vraw.f, SIG_SPEC) : inY;
        if (!smbOk)
            Asmb = 0.0f;
        float A = lerpf(Asmb, Avmb, amount);
        A *= confS;
        float q = lerpf(quality, 1.0f, amount);
        A *= lerpf(q, 1.0f, rcp_(1.0f + A));
        if (p.responsiveRoughnessThreshold > 0.0f) {
            float t = smoothstep01(g.roughness / p.responsiveRoughnessThreshold);
            A = fmin2(A, lerpf(p.responsiveMinAccum, p.maxASpec, t));
        }
        float nonLin = rcp_(1.0f + A);
        f4 hist = lerp4(smbHist, vmbHist, amount);
        float fastHist = lerpf(smbFast, vmbFast, amount);
        outw[sw] = pack_h4(lerp4(hist, in, nonLin));
        if (SH) {
            f4 in1 = unpack_h4(ctex[sw + S1]);
            f4 vmb1 = vmbOk ? blend4(vmb, vraw.t, sw + S1) : in1;
            outw[sw + S1] = pack_h4(lerp4(lerp4(smb1, vmb1, amount), in1, nonLin));
        }
        fastw |= (uint32_t)f2h(lerpf(fastHist, inY, rcp_(1.0f + fmin2(A, p.maxFastASpec)))) << (8 * lo);
        if (RELAX) {
            float m2 = inY * inY;
            float m2vmb = vmbOk ? blend1(vmb, vraw.m, SIG_SPEC) : m2;
            m2w |= (uint32_t)f2h(lerpf(lerpf(m2smb, m2vmb, amount), m2, nonLin)) << (8 * lo);
        }
        outSpecA = A;
        // bits 16..23: reprojection confidence of the specular history (RELAX A-trous edge-stopping relaxation)
        data2 |= (vmbBits << 4) | ((uint32_t)__builtin_floorf(fma_(sat(amount), 255.0f, 0.5f)) << 8) | (RELAX ? (uint32_t)__builtin_floorf(fma_(sat(q), 255.0f, 0.5f)) << 16 : 0u);
    }
    store_texel<RBPT>(p.tmp2, x, y, outw);
    store_luma(p.fast, x, y, LBPT, fastw);
    if (RELAX)
        store_luma(p.stab, x, y, LBPT, m2w);
    st<uint16_t>(p.data1Tmp, x, y, 2, pack_data1(outDiffA, outSpecA));
    st<uint32_t>(p.data2, x, y, 4, data2);
}

template <bool HAS_DIFF, bool HAS_SPEC, bool SH, bool RELAX>
__global__ __launch_bounds__(256) NRD_WAVES_PER_EU((SH || RELAX) ? NRD_TA_SH_WAVES : NRD_TA_WAVES) void k_temporal_accumulation(const ReblurParams p) {
    constexpr int NSIG = (HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0);
    constexpr int RBPT = (SH ? 16 : 8) * NSIG;
    const FrameConsts& c = p.c;
    int x, y, tx, ty, tflag;
    if (!my_pixel_w(c, x, y, tx, ty, tflag))
        return;
    // a tile without geometry: Tmp2, the fast history, Data1Tmp and Data2 of its pixels are read by nobody who has not tested the guide
    // first (HistoryFix skips the tile, its reconstruction taps and 5x5 windows test the tap's depth; next frame's footprints weigh a
    // texel beyond the range with an exact 0) - nothing to write (k_spatial has the full argument)
    if (NRD_SKIP_SKY_TILES && tile_is_sky(p, tx, ty, tflag))
        return;
    Guide g = decode_guide(ld_guide(p.guide, x, y), c.denoisingRange);
    if (g.sky) {
        ta_sky_stores<HAS_DIFF, HAS_SPEC, SH, RELAX>(p, x, y);
        return;
    }
    // ---- centre loads
    uint2 ctex[RBPT / 8];
    load_texel<RBPT>(p.tmp1, x, y, ctex);
    const float hitDist = HAS_SPEC ? h2f(ld<uint16_t>(p.hitTrack, x, y, 2)) : 0.0f;
#ifndef NRD_TA_SEQ_FOOTPRINTS // 1 (A/B): the stand-alone kernels fetch the virtual-motion footprint behind the surface-motion one too (round 5 measured
#define NRD_TA_SEQ_FOOTPRINTS 0 // it 0-6 % slower at 113 VGPRs = 4 waves; with NRD_TA_LEAN the SH flavours get below that)
#endif
    ta_pixel<HAS_DIFF, HAS_SPEC, SH, RELAX, NRD_TA_SEQ_FOOTPRINTS != 0>(p, x, y, g, ctex, hitDist, nullptr);
}

// PrePass + TemporalAccumulation of the REBLUR radiance flavours in ONE launch (spatial_pixel<..., FUSED>)
#ifndef NRD_FUSED_WAVES
#define NRD_FUSED_WAVES 5
#endif
template <bool HAS_DIFF, bool HAS_SPEC>
__global__ __launch_bounds__(256) NRD_WAVES_PER_EU(NRD_FUSED_WAVES) void k_prepass_temporal_accumulation(const ReblurParams p) {
    // (no arguments pinned at entry here: the kernel sits at its register limit - the orthographic flavour spilled 68 bytes with them, and
    // the perspective one gained nothing: profiles/r05_ab_pinned_arguments.txt)
    int x, y, tx, ty, tflag;
    if (!my_pixel(p.c, x, y, tx, ty, tflag, false))
        return;
    spatial_pixel<0, 0, HAS_DIFF, HAS_SPEC, true>(p, x, y, tx, ty, tflag);
}

// =====================================================================================================================
// 5x5 luma tile in LDS: 20x20 floats, NaN marks "sky / outside" (the consumer substitutes its own centre value)
// =====================================================================================================================
// (ring_pos - which thread fetches which position of the 2-texel ring of a 20x20 window - lives in nrd_device.h: SIGMA uses it too)

// `holes` is block-uniform: false when every texel of the staged 20x20 tile is valid (the common case), and the per-tap NaN test
// of the substitution is dropped - the sums are the same either way
NRD_DEV void moments5x5(const float* tile, int lx, int ly, float centre, bool holes, float& m1, float& m2) {
    m1 = 0.0f;
    m2 = 0.0f;
    if (holes) {
#pragma unroll
        for (int j = 0; j < 5; j++)
#pragma unroll
            for (int i = 0; i < 5; i++) {
                float f = tile[(ly + j) * 20 + lx + i];
                f = f != f ? centre : f;
                m1 += f;
                m2 = fma_(f, f, m2);
            }
    } else {
#pragma unroll
        for (int j = 0; j < 5; j++)
#pragma unroll
            for (int i = 0; i < 5; i++) {
                float f = tile[(ly + j) * 20 + lx + i];
                m1 += f;
                m2 = fma_(f, f, m2);
            }
    }
    m1 *= 1.0f / 25.0f;
    m2 *= 1.0f / 25.0f;
}

// =====================================================================================================================
// K4 HistoryFix
// =====================================================================================================================
template <bool HAS_DIFF, bool HAS_SPEC, bool SH>
__global__ __launch_bounds__(256) void k_history_fix(const ReblurParams p) {
    constexpr int NSIG = (HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0);
    constexpr int sb = SH ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    constexpr int RBPT = sb * NSIG;
    constexpr int LBPT = 2 * NSIG;
    constexpr int SIG_SPEC = HAS_DIFF ? 1 : 0;
    __shared__ float tile[NSIG][400];    // fast luma history
    __shared__ float tileCur[NSIG][400]; // incoming luma (anti-firefly only)
    const FrameConsts& c = p.c;
    if (NRD_PIN_ARGS)
        NRD_PIN_PLANES4(p.guide, p.data1Tmp, p.tmp2, p.fast);
    int tx, ty, tflag;
    if (!xcd_tile(c, tx, ty, tflag))
        return;
    // the centre loads of the pixel do not depend on the staged tiles: issued first, they travel with the staging loads (this
    // kernel spent 72 % of its wave time in s_waitcnt: staging round trip, barrier, then the centre round trip)
    int x = tx * 16 + (int)threadIdx.x, y = ty * 16 + (int)threadIdx.y;
    const bool live = x < c.W && y >= c.ownY0 && y < c.ownY1;
    const PlaneRef& outP = p.relax ? p.hist : p.tmp1; // RELAX: the fixed + clamped signal IS the next frame's history
    // REBLUR radiance flavours: the result goes out as tap texels {guide part | signal} for the Blur (nrd_device.h), one plane per signal
    const bool tapTex = !SH && p.tapTex != 0;
    auto sky_out = [&](const uint2 tg) {
        for (int sig = 0; sig < NSIG; sig++) {
            if (tapTex) {
                st<uint4>(p.tapA[(HAS_SPEC && sig == SIG_SPEC) ? 1 : 0], x, y, 16, uint4{tg.x, tg.y, 0u, 0u});
                continue;
            }
            st<uint2>(outP, x, y, RBPT, uint2{0u, 0u}, sig * sb);
            if (SH)
                st<uint2>(outP, x, y, RBPT, uint2{0u, 0u}, sig * sb + 8);
        }
        st<uint16_t>(p.data1, x, y, 2, (uint16_t)0);
    };
    // a tile without geometry (ClassifyTiles: Tiles = 1; block-uniform): every pixel takes the sky path - no staging, no barrier
    if (tile_is_sky(p, tx, ty, tflag)) {
        if (live)
            sky_out(tapTex ? ld_guide(p.guide, x, y) : uint2{0u, 0u});
        return;
    }
    const int cxp = imin(x, c.W - 1), cyp = imin(imax(y, 0), c.resH - 1); // clamped: threads outside still take part in the staging
    const uint2 graw = ld_guide(p.guide, cxp, cyp);
    const uint16_t data1Raw = ld<uint16_t>(p.data1Tmp, cxp, cyp, 2);
    uint2 ctex[RBPT / 8];
    load_texel<RBPT>(p.tmp2, cxp, cyp, ctex);
    const uint32_t fastRaw = p.clampEnabled ? load_luma(p.fast, cxp, cyp, LBPT) : 0u;
    bool holes = false; // some texel of the staged tiles is sky / outside (block-uniform)
    if (p.clampEnabled || p.antiFirefly) {
        // 20x20 luma tiles of all signals: the interior from the centre loads, the ring as depth + one luma texel per position
        // (clamped unconditional loads)
        int tid = (int)threadIdx.y * 16 + (int)threadIdx.x;
        int bad = 0;
        {
            const int i = ((int)threadIdx.y + 2) * 20 + (int)threadIdx.x + 2;
            const int gy = y + c.yOff;
            bool ok = x < c.W && gy >= 0 && gy < c.H && y >= 0 && y < c.resH && absf(u2f(graw.x)) <= c.denoisingRange;
            bad |= ok ? 0 : 1;
            const uint32_t l = p.clampEnabled ? fastRaw : load_luma(p.fast, cxp, cyp, LBPT);
#pragma unroll
            for (int sig = 0; sig < NSIG; sig++)
                tile[sig][i] = ok ? h2f((uint16_t)(l >> (16 * sig))) : u2f(0x7fc00000u);
            if (p.antiFirefly) {
#pragma unroll
                for (int sig = 0; sig < NSIG; sig++)
                    tileCur[sig][i] = ok ? texel_luma(ctex[sig * (sb / 8)], p.relax != 0) : u2f(0x7fc00000u);
            }
        }
        int lx, ly;
        if (ring_pos(tid, lx, ly)) {
            const int i = ly * 20 + lx;
            int px = tx * 16 + lx - 2, py = ty * 16 + ly - 2, gy = py + c.yOff;
            bool inside = px >= 0 && px < c.W && gy >= 0 && gy < c.H && py >= 0 && py < c.resH;
            int cx = imin(imax(px, 0), c.W - 1), cy = imin(imax(py, 0), c.resH - 1);
            float zt = ld<float>(p.guide, cx, cy, GUIDE_BYTES, 0);
            uint32_t l = load_luma(p.fast, cx, cy, LBPT);
            bool ok = inside && absf(zt) <= c.denoisingRange;
            bad |= ok ? 0 : 1;
#pragma unroll
            for (int sig = 0; sig < NSIG; sig++)
                tile[sig][i] = ok ? h2f((uint16_t)(l >> (16 * sig))) : u2f(0x7fc00000u);
            if (p.antiFirefly) {
#pragma unroll
                for (int sig = 0; sig < NSIG; sig++)
                    tileCur[sig][i] = ok ? ((RELAX_LINEAR_RGB && p.relax) ? texel_luma(ld<uint2>(p.tmp2, cx, cy, RBPT, sig * sb), true) : h2f(ld<uint16_t>(p.tmp2, cx, cy, RBPT, sig * sb)))
                                         : u2f(0x7fc00000u);
            }
        }
        holes = __syncthreads_or(bad) != 0;
    }
    if (!live)
        return;
    Guide g = decode_guide(graw, c.denoisingRange);
    const uint2 tapGuide = graw; // the tap texels carry the pixel's guide texel as it is
    if (g.sky) {
        sky_out(tapGuide);
        return;
    }
    const int gy0 = y + c.yOff;
    float A[2];
    unpack_data1(data1Raw, A[0], A[1]);
    float outA[2] = {A[0], A[1]};
    bool geoReady = false;
    PixelGeo pg;
#pragma unroll
    for (int sig = 0; sig < NSIG; sig++) {
        const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
        const int ai = isSpec ? 1 : 0;
        float rough = isSpec ? g.roughness : 1.0f;
        uint32_t minMat = isSpec ? p.minMatSpec : p.minMatDiff;
        f4 val = unpack_h4(ctex[sig * (sb / 8)]);
        f4 val1 = SH ? unpack_h4(ctex[sig * (sb / 8) + 1]) : f4{0, 0, 0, 0};
        float Acur = A[ai];
        if (Acur < (float)p.historyFixFrameNum && p.historyFixFrameNum > 0) {
            float normA = sat(Acur / (float)p.historyFixFrameNum);
            int stride = (int)__builtin_floorf(fma_((float)p.historyFixStride, 1.0f - normA, 0.5f));
            if (stride > 0) {
                if (!geoReady) {
                    pg = pixel_geo(c, g, x, gy0, p.planeDistanceSensitivity);
                    geoReady = true;
                }
                float angle = spec_lobe_half_angle(rough) * lerpf(p.lobeAngleFraction, 1.0f, rcp_(1.0f + Acur));
                float normalW = rcp_(fmax2(angle, NORMAL_ANGLE_MIN));
                normalW *= strand_normal_relax(c, g.mat, absf(g.z)); // CommonSettings::strandMaterialID: thin strands relax the normal test
                float normalW2 = nw_param(normalW);
                float roughA = rcp_(lerpf(0.01f, 1.0f, sat(rough * p.roughnessFraction)));
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
                        Guide gs = decode_guide(ld_guide(p.guide, px, py), c.denoisingRange);
                        if (gs.sky || material_mismatch(g.mat, gs.mat, minMat))
                            continue;
                        float w = rcp_(1.0f + (float)(i * i + j * j));
                        w *= geo_weight(pg, (float)px, (float)gy, gs.z);
                        w *= p.relax ? pow01(normal_cos(normal_codes(g.nw), gs.nw), p.hfNormalPower) : normal_weight(normal_dist2(normal_codes(g.nw), gs.nw), normalW2);
                        if (isSpec)
                            w *= smoothstep01(1.0f - absf(fma_(gs.roughness, roughA, roughB)));
                        float tA[2];
                        unpack_data1(ld<uint16_t>(p.data1Tmp, px, py, 2), tA[0], tA[1]);
                        w *= 1.0f + tA[ai];
                        sum = fma4(unpack_h4(ld<uint2>(p.tmp2, px, py, RBPT, sig * sb)), w, sum);
                        if (SH)
                            sum1 = fma4(unpack_h4(ld<uint2>(p.tmp2, px, py, RBPT, sig * sb + 8)), w, sum1);
                        wsum += w;
                    }
                val = mul4(sum, rcp_(wsum));
                val1 = mul4(sum1, rcp_(wsum));
            }
        }
        if (p.clampEnabled) {
            float fc = h2f((uint16_t)(fastRaw >> (16 * sig)));
            float m1, m2;
            moments5x5(tile[sig], (int)threadIdx.x, (int)threadIdx.y, fc, holes, m1, m2);
            float sigma = sqrt_(fmax2(fma_(-m1, m1, m2), 0.0f)) * p.fastHistoryClampingSigmaScale;
            float Y = signal_luma(val, p.relax != 0);
            float Yc = clampf(Y, m1 - sigma, m1 + sigma);
            float scale = (Yc + 1e-6f) * rcps_(Y + 1e-6f);
            clamp_luma(val, Yc, scale, p.relax != 0);
            val1.x *= scale;
            val1.y *= scale;
            val1.z *= scale;
            float f = sat(absf(Yc - Y) * rcp_(fmax2(fmax2(Y, Yc), 1e-6f)));
            if (p.relax)
                f *= p.alAccel;
            outA[ai] = lerpf(Acur, fmin2(Acur, isSpec ? p.maxFastASpec : p.maxFastA), f);
            if (p.relax) { // antilag: histories far from the fast 5x5 mean (in spatial + temporal sigmas) are reset
                float sigS = sqrt_(fmax2(fma_(-m1, m1, m2), 0.0f));
                float sigT = sqrt_(fmax2(fma_(-Y, Y, h2f(ld<uint16_t>(p.stab, x, y, LBPT, sig * 2))), 0.0f));
                float thr = fma_(p.alSpatial, sigS, p.alTemporal * sigT);
                float over = sat(fma_(absf(Y - m1), rcp_(fmax2(thr, 1e-6f)), -1.0f));
                outA[ai] *= fma_(-p.alReset, over, 1.0f);
            }
        }
        if (p.antiFirefly) { // luma clamped to the centre-less 5x5 moments of the incoming signal
            float cc = tileCur[sig][((int)threadIdx.y + 2) * 20 + (int)threadIdx.x + 2];
            float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 5; j++)
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    if (i == 2 && j == 2)
                        continue;
                    float f = tileCur[sig][((int)threadIdx.y + j) * 20 + (int)threadIdx.x + i];
                    f = f != f ? cc : f;
                    m1 += f;
                    m2 = fma_(f, f, m2);
                }
            m1 *= 1.0f / 24.0f;
            m2 *= 1.0f / 24.0f;
            float sigma = sqrt_(fmax2(fma_(-m1, m1, m2), 0.0f)) * p.fireflyScale;
            float Y = signal_luma(val, p.relax != 0);
            float Yc = clampf(Y, m1 - sigma, m1 + sigma);
            float scale = (Yc + 1e-6f) * rcps_(Y + 1e-6f);
            clamp_luma(val, Yc, scale, p.relax != 0);
            val1.x *= scale;
            val1.y *= scale;
            val1.z *= scale;
        }
        if (tapTex) {
            const uint2 sv = pack_h4(val);
            st<uint4>(p.tapA[isSpec ? 1 : 0], x, y, 16, uint4{tapGuide.x, tapGuide.y, sv.x, sv.y});
        } else
            st<uint2>(outP, x, y, RBPT, pack_h4(val), sig * sb);
        if (SH)
            st<uint2>(outP, x, y, RBPT, pack_h4(val1), sig * sb + 8);
    }
    st<uint16_t>(p.data1, x, y, 2, pack_data1(outA[0], outA[1]));
}

// =====================================================================================================================
// K7 TemporalStabilization (+ split screen)
// =====================================================================================================================
// stabilized-luma history of one footprint: the four texels (all signals) are fetched up front, then blended with the
// validity bits TemporalAccumulation recorded for that footprint
NRD_DEV void load_stab(const ReblurParams& p, const FootPos& fp, int lbpt, uint32_t (&raw)[4]) {
    const FrameConsts& c = p.c;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int tx = imin(imax(fp.ix + (i & 1), 0), c.Wprev - 1), ty = imin(imax(fp.iy + (i >> 1) - c.yOff, 0), c.resH - 1);
        raw[i] = load_luma(p.stabPrev, tx, ty, lbpt);
    }
}
NRD_DEV bool blend_stab(const FootPos& fp, const uint32_t (&raw)[4], int half, uint32_t bits, float& out) {
    float fx = fp.fx, fy = fp.fy;
    float bw[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
    float sum = 0.0f, wsum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        bool on = (bits & (1u << i)) != 0u;
        float acc = fma_(h2f((uint16_t)(raw[i] >> (16 * half))), bw[i], sum);
        sum = on ? acc : sum;
        wsum = on ? wsum + bw[i] : wsum;
    }
    out = sum * (rcp_(wsum));
    return fp.sane && wsum > 0.0f;
}

#ifndef NRD_TS_WAVES // waves per SIMD the register allocator aims for in TemporalStabilization (1: no bound). Unbounded the
#define NRD_TS_WAVES 8 // kernel takes 49 VGPRs but 106 SGPRs = 7 waves; held to 8 waves' budget: 0.1128 -> 0.1084 ms at 4K (profiles/r05_ab_ts_waves_ct_tiles.txt)
#endif
template <bool HAS_DIFF, bool HAS_SPEC, bool SH>
__global__ __launch_bounds__(256) NRD_WAVES_PER_EU(NRD_TS_WAVES) void k_temporal_stabilization(const ReblurParams p) {
    constexpr int NSIG = (HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0);
    constexpr int sb = SH ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    constexpr int RBPT = sb * NSIG;
    constexpr int SW = sb / 8, S1 = SH ? 1 : 0;
    constexpr int LBPT = 2 * NSIG;
    constexpr int SIG_SPEC = HAS_DIFF ? 1 : 0;
    __shared__ float tile[NSIG][400];
    const FrameConsts& c = p.c;
    if (NRD_PIN_ARGS)
        NRD_PIN_PLANES4(p.guide, p.hist, p.inMV, p.data2);
    int tx, ty, tflag;
    if (!xcd_tile(c, tx, ty, tflag))
        return;
    // Memory round trips are what this kernel waits for (64 % of its wave time sat in s_waitcnt): the centre loads of the pixel do
    // not depend on the staged tile, so they are issued FIRST and travel together with the staging loads - one round trip, then
    // the barrier; the history footprints (which need the centre's motion vector) travel while the 5x5 moments are read from LDS.
    int x = tx * 16 + (int)threadIdx.x, y = ty * 16 + (int)threadIdx.y;
    const bool live = x < c.W && y >= c.ownY0 && y < c.ownY1;
    auto sky_out = [&]() {
        const bool split = ((float)x + 0.5f) * c.invW < c.splitScreen;
#pragma unroll
        for (int sig = 0; sig < NSIG; sig++) {
            const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
            const PlaneRef& o = isSpec ? p.outSpec : p.outDiff;
            const PlaneRef& in = isSpec ? p.inSpec : p.inDiff;
            // split screen shows the noisy input: the slot itself, or its dense PrepareInputs copy when that pass ran
            if (SH && p.dirOcc) // single {SH1.xyz, SH0.x} texel out
                st<uint2>(o, x, y, 8, split ? pack_dir(p, dir_pass(p, x, y)) : uint2{0u, 0u});
            else
                store_signal(p, o, x, y, split ? load_signal(p, in, x, y, 8, 0, p.occlusion != 0 && !p.prepared) : f4{0, 0, 0, 0});
            if (SH && !p.dirOcc)
                st<uint2>(isSpec ? p.outSpec1 : p.outDiff1, x, y, 8, split ? pack_h4(unpack_h4(ld<uint2>(isSpec ? p.inSpec1 : p.inDiff1, x, y, 8))) : uint2{0u, 0u});
            st<uint16_t>(p.stab, x, y, LBPT, (uint16_t)0, sig * 2);
        }
    };
    // a tile without geometry (ClassifyTiles: Tiles = 1; block-uniform): every pixel takes the sky path - no staging, no barrier
    if (tile_is_sky(p, tx, ty, tflag)) {
        if (live)
            sky_out();
        return;
    }
    const int cxp = imin(x, c.W - 1), cyp = imin(imax(y, 0), c.resH - 1); // clamped: threads outside still take part in the staging
    const uint2 graw = ld_guide(p.guide, cxp, cyp);
    uint2 ctex[RBPT / 8];
    load_texel<RBPT>(p.hist, cxp, cyp, ctex);
    const uint2 mvTexel = ld<uint2>(p.inMV, cxp, cyp, 8);
    const uint32_t data2 = ld<uint32_t>(p.data2, cxp, cyp, 4);
    const uint16_t data1Raw = ld<uint16_t>(p.data1, cxp, cyp, 2);
    const uint16_t hitRaw = HAS_SPEC ? ld<uint16_t>(p.hitTrack, cxp, cyp, 2) : (uint16_t)0;
    // 20x20 luma tiles of all signals: the interior positions are the threads' own pixels (centre loads above), threads 0..143 fetch
    // one position of the 2-texel ring each (guide depth + radiance texel, unconditionally at clamped coordinates; NaN marks "sky /
    // outside"). The ring loads go to registers first: the history footprints below only need the centre loads (issued earlier, so
    // they return earlier) and are issued BEFORE the ring texels are waited for and written to LDS - centre, ring and footprint
    // traffic overlap instead of forming three dependent round trips.
    const int tid = (int)threadIdx.y * 16 + (int)threadIdx.x;
    int rlx = 0, rly = 0;
    const bool ringOn = ring_pos(tid, rlx, rly);
    const int ringI = rly * 20 + rlx;
    float ringZ;
    uint2 ringT[RBPT / 8];
    bool ringIn;
    {
        int px = tx * 16 + rlx - 2, py = ty * 16 + rly - 2, gy = py + c.yOff;
        ringIn = px >= 0 && px < c.W && gy >= 0 && gy < c.H && py >= 0 && py < c.resH;
        int cx = imin(imax(px, 0), c.W - 1), cy = imin(imax(py, 0), c.resH - 1);
        ringZ = ld<float>(p.guide, cx, cy, GUIDE_BYTES, 0);
        load_texel<RBPT>(p.hist, cx, cy, ringT);
    }
    const int gy0 = y + c.yOff;
    float u = ((float)x + 0.5f) * c.invW, v = ((float)gy0 + 0.5f) * c.invH;
    bool split = u < c.splitScreen;
    Guide g = decode_guide(graw, c.denoisingRange);
    // ---- both history footprints (positions from the centre's motion vector; a sky / outside pixel computes harmless clamped ones)
    f4 mvRaw = unpack_h4(mvTexel);
    float A[2];
    unpack_data1(data1Raw, A[0], A[1]);
    float hitDist = HAS_SPEC ? h2f(hitRaw) : 0.0f;
    f3 Xv = reconstruct_px(c.pv, (float)x, (float)gy0, g.z);
    Reproj r = reproject(c, Xv, u, v, mvRaw);
    const bool historyOk = c.historyOk != 0;
    float amount = (float)((data2 >> 8) & 255u) * (1.0f / 255.0f);
    FootPos spos = foot_pos(c, r.su, r.sv);
    uint32_t sraw[4], vraw[4] = {0u, 0u, 0u, 0u};
    load_stab(p, spos, LBPT, sraw);
    FootPos vpos = spos;
    if (HAS_SPEC) {
        float tu, tv;
        // (the dominant factor is EVALUATED here: from the table it would be one more dependent round trip in front of the virtual footprint's
        // loads in a kernel that waits for memory, not for issue slots - NRD_ROUGH_LUT_TS 1 is the A/B switch)
#ifndef NRD_ROUGH_LUT_TS
#define NRD_ROUGH_LUT_TS 0
#endif
        bool vOk = virtual_uv(c, r, hitDist, NRD_ROUGH_LUT_TS ? dominant_factor_of(p, g) : spec_dominant_factor(g.roughness), g.mat, tu, tv) && amount > 0.0f;
        vpos = foot_pos(c, vOk ? tu : -10.0f, vOk ? tv : -10.0f); // unusable virtual position: lands outside, never validates
        load_stab(p, vpos, LBPT, vraw);
    }
    __builtin_amdgcn_sched_barrier(0);
    int bad = 0;
    {
        const bool ok = x < c.W && gy0 >= 0 && gy0 < c.H && y >= 0 && y < c.resH && absf(u2f(graw.x)) <= c.denoisingRange;
        bad |= ok ? 0 : 1;
        const int i = ((int)threadIdx.y + 2) * 20 + (int)threadIdx.x + 2;
#pragma unroll
        for (int sig = 0; sig < NSIG; sig++)
            tile[sig][i] = ok ? h2f((uint16_t)ctex[sig * SW].x) : u2f(0x7fc00000u);
    }
    if (ringOn) {
        const bool ok = ringIn && absf(ringZ) <= c.denoisingRange;
        bad |= ok ? 0 : 1;
#pragma unroll
        for (int sig = 0; sig < NSIG; sig++)
            tile[sig][ringI] = ok ? h2f((uint16_t)ringT[sig * SW].x) : u2f(0x7fc00000u);
    }
    const bool holes = __syncthreads_or(bad) != 0; // some texel of the staged tiles is sky / outside (block-uniform)
    if (!live)
        return;
    if (g.sky) {
        sky_out();
        return;
    }
    // the 5x5 moments come from LDS: computed while the footprints travel
    float m1s[NSIG], m2s[NSIG];
#pragma unroll
    for (int sig = 0; sig < NSIG; sig++)
        moments5x5(tile[sig], (int)threadIdx.x, (int)threadIdx.y, h2f((uint16_t)ctex[sig * SW].x), holes, m1s[sig], m2s[sig]);
    __builtin_amdgcn_sched_barrier(0);
    uint32_t stabw = 0u; // stabilized luma of the pixel's signals: one texel, one store
#pragma unroll
    for (int sig = 0; sig < NSIG; sig++) {
        const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
        f4 cur = unpack_h4(ctex[sig * SW]);
        const float m1 = m1s[sig], m2 = m2s[sig];
        float sigma = sqrt_(fmax2(fma_(-m1, m1, m2), 0.0f));
        float smbY, vmbY = 0.0f;
        bool smbOk = blend_stab(spos, sraw, sig, data2 & 15u, smbY) && historyOk;
        bool vmbOk = false;
        if (isSpec)
            vmbOk = blend_stab(vpos, vraw, sig, (data2 >> 4) & 15u, vmbY) && historyOk;
        float Yhist = (smbOk && vmbOk) ? lerpf(smbY, vmbY, amount) : (smbOk ? smbY : (vmbOk ? vmbY : cur.x));
        bool have = smbOk || vmbOk;
        float Acur = A[isSpec ? 1 : 0];
        float Y = cur.x;
        float band = sigma * p.antilagSigmaScale;
        float dlt = fmax2(absf(Yhist - m1) - band, 0.0f) * rcps_(fmax2(Yhist, m1) + 1e-6f);
        float antilag = rcps_(fma_(dlt * p.antilagSensitivity, Acur, 1.0f));
        float Yclamped = clampf(Yhist, m1 - band, m1 + band);
        float stabFrames = have ? fmin2(Acur, p.maxStab) * antilag : 0.0f;
        float wHist = stabFrames * rcp_(1.0f + stabFrames);
        float Yout = lerpf(Y, Yclamped, wHist);
        float scale = (Yout + 1e-6f) * rcps_(Y + 1e-6f);
        f4 o = {Yout, cur.y * scale, cur.z * scale, cur.w};
        if (p.returnHistLen) // OCCLUSION variants: the single output channel reports the normalised history length instead
            o.x = sat(Acur * p.invMaxA);
        stabw |= (uint32_t)f2h(Yout) << (16 * sig);
        const PlaneRef& op = isSpec ? p.outSpec : p.outDiff;
        const PlaneRef& in = isSpec ? p.inSpec : p.inDiff;
        if (SH && p.dirOcc) {
            f4 c1 = unpack_h4(ctex[sig * SW + S1]);
            st<uint2>(op, x, y, 8, pack_dir(p, split ? dir_pass(p, x, y) : f4{c1.x * scale, c1.y * scale, c1.z * scale, Yout}));
            continue;
        }
        store_signal(p, op, x, y, split ? load_signal(p, in, x, y, 8, 0, p.occlusion != 0 && !p.prepared) : o);
        if (SH) {
            f4 c1 = unpack_h4(ctex[sig * SW + S1]);
            f4 o1 = {c1.x * scale, c1.y * scale, c1.z * scale, c1.w};
            st<uint2>(isSpec ? p.outSpec1 : p.outDiff1, x, y, 8, split ? pack_h4(unpack_h4(ld<uint2>(isSpec ? p.inSpec1 : p.inDiff1, x, y, 8))) : pack_h4(o1));
        }
    }
    store_luma(p.stab, x, y, LBPT, stabw);
}

// =====================================================================================================================
// Validation overlay (CommonSettings::enableValidation, OUT_VALIDATION bound at Source/NRDSample.cpp:452, RGBA8): per pixel
// {diffuse accumulated frames / 63, specular accumulated frames / 63, |viewZ| / denoisingRange, virtual-motion amount}; 0 on sky
// =====================================================================================================================
__global__ __launch_bounds__(256) void k_validation(const ReblurParams p) {
    const FrameConsts& c = p.c;
    int x, y, tx, ty;
    if (!my_pixel(c, x, y, tx, ty))
        return;
    Guide g = decode_guide(ld_guide(p.guide, x, y), c.denoisingRange);
    uint32_t packed = 0;
    if (!g.sky) {
        float dA, sA;
        unpack_data1(ld<uint16_t>(p.data1, x, y, 2), dA, sA);
        uint32_t r = (uint32_t)__builtin_floorf(fma_(sat(dA * (1.0f / 63.0f)), 255.0f, 0.5f));
        uint32_t gg = (uint32_t)__builtin_floorf(fma_(sat(sA * (1.0f / 63.0f)), 255.0f, 0.5f));
        uint32_t b = (uint32_t)__builtin_floorf(fma_(sat(absf(g.z) * rcp_(c.denoisingRange)), 255.0f, 0.5f));
        uint32_t a = (ld<uint32_t>(p.data2, x, y, 4) >> 8) & 255u;
        packed = r | (gg << 8) | (b << 16) | (a << 24);
    }
    st<uint32_t>(p.outValidation, x, y, 4, packed);
}

// =====================================================================================================================
// RELAX A-trous iteration: variance-guided 3x3 at stride 2^it (Source/NRDSample.cpp:1642-1657 feeds the settings; outputs are
// decoded by RELAX_BackEnd_UnpackRadiance, Shaders/Composition.cs.hlsl:160-161). All lanes use the same tap offsets, so the
// gathers of a 16x4-pixel wave are 16x4-texel groups: fully coalesced at every stride; neighbouring tiles share taps in L2.
// =====================================================================================================================
// FIRST: iteration 0 (variance from the luminance moments, 3x3 spatial estimate for short histories). The 3x3 taps of the
// signals share their positions, so the tap loop is the OUTER loop: one guide gather + decode + plane/normal terms per tap
// serve both signals (each signal still sees exactly the operation sequence of the oracle).
// LS > 0 (strides 1, 2, 4 = iterations 0..2): the workgroup's (16 + 2 LS)^2 window of guide + radiance (+ moment) texels is staged
// in LDS once - every texel is read by up to 9 pixels - and the taps become LDS reads at compile-time offsets: no per-tap
// address arithmetic, no bounds tests (a texel outside the frame / the held rows is staged with viewZ = NaN, i.e. as sky).
// LS = 0 (strides >= 8, window too large for LDS at a useful occupancy): coalesced global gathers, batched.
#ifndef NRD_ATROUS_SIGNAL_ROUNDS // A/B switch: see k_relax_atrous
#define NRD_ATROUS_SIGNAL_ROUNDS 0
#endif
#define NRD_ATROUS_SIGNAL_ROUNDS_ON (NRD_ATROUS_SIGNAL_ROUNDS != 0)
#ifndef NRD_ATROUS_WAVES // waves per SIMD the register allocator aims for in the iterations behind the first (the first holds the moments too: 3)
#define NRD_ATROUS_WAVES 4
#endif
template <bool HAS_DIFF, bool HAS_SPEC, bool SH, bool FIRST, int LS>
__global__ __launch_bounds__(256) NRD_WAVES_PER_EU(FIRST ? 1 : NRD_ATROUS_WAVES) void k_relax_atrous(const AtrousParams p) {
    constexpr int NSIG = (HAS_DIFF ? 1 : 0) + (HAS_SPEC ? 1 : 0);
    constexpr int sb = SH ? 16 : 8; // bytes per signal in the radiance texels: SH0 (+ SH1 at +8 in SH mode)
    constexpr int RBPT = sb * NSIG;
    constexpr int SW = sb / 8; // uint2 words per signal
    constexpr int TW = RBPT / 8; // uint2 words per radiance texel
    constexpr int LBPT = 2 * NSIG;
    constexpr int SIG_SPEC = HAS_DIFF ? 1 : 0;
    constexpr int T = 16 + 2 * LS; // staged window edge
    __shared__ uint2 sG[LS ? T * T : 1];
    __shared__ uint2 sT[LS ? T * T * TW : 1];
    __shared__ uint32_t sM[(LS && FIRST) ? T * T : 1];
    const FrameConsts& c = p.c;
    // rows / columns a tap may land on: inside the frame and inside the rows this instance holds
    const int loY = imax(0, -c.yOff), hiY = imin(c.resH, c.H - c.yOff) - 1;
    int x, y, tx, ty, tflag;
    uint16_t d1raw = 0;
    uint32_t d2raw = 0;
    const bool last = p.last != 0;
    auto sky_out = [&]() { // what a sky pixel stores
        const bool split = last && ((float)x + 0.5f) * c.invW < c.splitScreen;
#pragma unroll
        for (int sig = 0; sig < NSIG; sig++) {
            const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
            if (last) {
                const PlaneRef& o = isSpec ? p.outSpec : p.outDiff;
                const PlaneRef& in = isSpec ? p.inSpec : p.inDiff;
                st<uint2>(o, x, y, 8, split ? pack_h4(unpack_h4(ld<uint2>(in, x, y, 8))) : uint2{0u, 0u});
                if (SH)
                    st<uint2>(isSpec ? p.outSpec1 : p.outDiff1, x, y, 8, split ? pack_h4(unpack_h4(ld<uint2>(isSpec ? p.inSpec1 : p.inDiff1, x, y, 8))) : uint2{0u, 0u});
            } else {
                st<uint2>(p.out, x, y, RBPT, uint2{0u, 0u}, sig * sb);
                if (SH)
                    st<uint2>(p.out, x, y, RBPT, uint2{0u, 0u}, sig * sb + 8);
            }
        }
    };
    if (LS) {
        if (!xcd_tile(c, tx, ty, tflag))
            return;
        if (tile_is_sky(p.tiles, tx, ty, tflag)) { // a tile without geometry (block-uniform): the per-pixel sky stores, no staging, no barrier
            x = tx * 16 + (int)threadIdx.x;
            y = ty * 16 + (int)threadIdx.y;
            if (x < c.W && y >= c.ownY0 && y < c.ownY1)
                sky_out();
            return;
        }
        // ALL staging loads of this thread go out before the first LDS write (one memory round trip for the window instead of one
        // per 256 texels), together with the pixel's own data1 / data2 words
        const int tid = (int)threadIdx.y * 16 + (int)threadIdx.x;
        x = tx * 16 + (int)threadIdx.x;
        y = ty * 16 + (int)threadIdx.y;
        const int sx = imin(x, c.W - 1), sy = imin(y, c.resH - 1);
        if (FIRST)
            d1raw = ld<uint16_t>(p.data1, sx, sy, 2);
        if (HAS_SPEC)
            d2raw = ld<uint32_t>(p.data2, sx, sy, 4);
        constexpr int TRIPS = (T * T + 255) / 256;
        uint2 gq[TRIPS];
        uint2 tq[TRIPS][TW];
        uint32_t mq[TRIPS];
        bool ins[TRIPS];
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            const int i = imin(tid + 256 * k, T * T - 1); // (the last trip's surplus threads re-load the last texel and do not store it)
            const int lx = i % T, ly = i / T;
            const int px = tx * 16 + lx - LS, py = ty * 16 + ly - LS;
            ins[k] = ((uint32_t)px < (uint32_t)c.W) & ((uint32_t)(py - loY) <= (uint32_t)(hiY - loY));
            const int cpx = imin(imax(px, 0), c.W - 1), cpy = imin(imax(py, loY), hiY);
            gq[k] = ld_guide(p.guide, cpx, cpy);
            load_texel<RBPT>(p.in, cpx, cpy, tq[k]);
            mq[k] = FIRST ? load_luma(p.mom, cpx, cpy, LBPT) : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            const int i = tid + 256 * k;
            if (i < T * T) {
                gq[k].x = ins[k] ? gq[k].x : 0x7fc00000u;
                sG[i] = gq[k];
#pragma unroll
                for (int w = 0; w < TW; w++)
                    sT[i * TW + w] = tq[k][w];
                if (FIRST)
                    sM[i] = mq[k];
            }
        }
        __syncthreads();
        if (!(x < c.W && y >= c.ownY0 && y < c.ownY1))
            return;
    } else if (!my_pixel_w(c, x, y, tx, ty)) // no barrier in the gather flavour: one wave per workgroup
        return;
    const int ci = ((int)threadIdx.y + LS) * T + (int)threadIdx.x + LS; // this pixel in the staged window
    const int it = p.it;
    const int stride = LS ? LS : 1 << it;
    const int gy0 = y + c.yOff;
    float u = ((float)x + 0.5f) * c.invW;
    bool split = last && u < c.splitScreen;
    Guide g = decode_guide(LS ? sG[ci] : ld_guide(p.guide, x, y), c.denoisingRange);
    if (g.sky) {
        sky_out();
        return;
    }
    PixelGeo pg = pixel_geo(c, g, x, gy0, p.depthSens);
    float A[2] = {0.0f, 0.0f};
    if (FIRST)
        unpack_data1(LS ? d1raw : ld<uint16_t>(p.data1, x, y, 2), A[0], A[1]);
    uint2 ctex[RBPT / 8];
    if (LS) {
#pragma unroll
        for (int w = 0; w < TW; w++)
            ctex[w] = sT[ci * TW + w];
    } else
        load_texel<RBPT>(p.in, x, y, ctex);
    f4 c0[NSIG], sum1[NSIG];
    float c0Y[NSIG]; // luminance of the centre texel
    f3 sum[NSIG];
    float sumVar[NSIG], wsum[NSIG], invL[NSIG], normalW2[NSIG], minLw[NSIG];
    uint32_t matFloor[NSIG], matClass[NSIG]; // material test of the taps (nrd_device.h material_class)
    float roughA = 0.0f, roughB = 0.0f, roughRelax = 1.0f;
    // everything a signal's taps need (run for both signals in front of the tap loop - or, NRD_ATROUS_SIGNAL_ROUNDS, in front of its own round)
    auto setup_signal = [&](const int sig) {
        const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
        const int si = isSpec ? 1 : 0;
        float rough = isSpec ? g.roughness : 1.0f;
        matFloor[sig] = material_floor(isSpec ? p.minMatSpec : p.minMatDiff);
        matClass[sig] = material_class(g.mat, matFloor[sig]);
        minLw[sig] = p.minLw[si];
        c0[sig] = unpack_h4(ctex[sig * SW]);
        c0Y[sig] = signal_luma(c0[sig], true);
        sum1[sig] = SH ? unpack_h4(ctex[sig * SW + (SH ? 1 : 0)]) : f4{0, 0, 0, 0};
        float var;
        if (FIRST) {
            float m2 = h2f(LS ? (uint16_t)(sM[ci] >> (16 * sig)) : ld<uint16_t>(p.mom, x, y, LBPT, sig * 2));
            var = fmax2(fma_(-c0Y[sig], c0Y[sig], m2), 0.0f);
            if (A[si] < p.histThreshold) { // short history: 3x3 spatial estimate
                float sy = 0.0f, sy2 = 0.0f, n = 0.0f;
                for (int j = -1; j <= 1; j++)
                    for (int i = -1; i <= 1; i++) {
                        float Y;
                        if (LS) { // iteration 0 filters the history itself: the staged window holds these texels
                            int q = ci + j * T + i;
                            if (!(absf(u2f(sG[q].x)) <= c.denoisingRange))
                                continue;
                            Y = texel_luma(sT[q * TW + sig * SW], true);
                        } else {
                            int px = x + i, py = y + j, gy = py + c.yOff;
                            if (px < 0 || px >= c.W || gy < 0 || gy >= c.H || py < 0 || py >= c.resH)
                                continue;
                            if (!(absf(ld<float>(p.guide, px, py, GUIDE_BYTES, 0)) <= c.denoisingRange))
                                continue;
                            Y = RELAX_LINEAR_RGB ? texel_luma(ld<uint2>(p.hist, px, py, RBPT, sig * sb), true) : h2f(ld<uint16_t>(p.hist, px, py, RBPT, sig * sb));
                        }
                        sy += Y;
                        sy2 = fma_(Y, Y, sy2);
                        n += 1.0f;
                    }
                float inv = rcp_(n);
                float my = sy * inv;
                var = fmax2(var, fmax2(fma_(-my, my, sy2 * inv), 0.0f));
            }
            if (isSpec)
                var = fma_(var, p.specularVarianceBoost, var);
        } else
            var = c0[sig].w;
        float sigma = wsqrt_(var);
        invL[sig] = 0.3333f * wrcp_(fma_(p.phi[si], sigma, 1e-4f));
        // (specular: the arctangent of the lobe half angle hangs on the roughness code alone - the table ClassifyTiles wrote, entry word 2)
        float angle = ((NRD_ROUGH_LUT && isSpec) ? p.roughLut[(f2u(g.z) & 1023u) * 4u + 2u] : spec_lobe_half_angle(rough)) * p.lobeAngleFraction;
        if (isSpec)
            angle += p.lobeSlack;
        float normalW = wrcp_(fmax2(angle, NORMAL_ANGLE_MIN));
        normalW *= strand_normal_relax(c, g.mat, absf(g.z)); // CommonSettings::strandMaterialID: thin strands relax the normal test
        if (p.confDriven) { // confidenceDriven*: low history confidence relaxes the luminance / normal edge stopping of both signals
            float conf = sample_confidence(isSpec ? p.confS : p.confD, u, ((float)gy0 + 0.5f) * c.invH);
            float cd = sat(p.confMult * (1.0f - conf));
            invL[sig] *= fma_(-cd, p.confLumRelax, 1.0f);
            normalW *= fma_(-cd, p.confNormRelax, 1.0f);
        }
        if (LS && isSpec) { // fine iterations: relax the edge stopping where the specular history was reprojected with low confidence
            float conf = (float)((d2raw >> 16) & 255u) * (1.0f / 255.0f);
            invL[sig] *= lerpf(1.0f, conf, p.lumRelax);
            normalW *= lerpf(1.0f, conf, p.normRelax);
            roughRelax = lerpf(1.0f, conf, p.roughRelax);
        }
        normalW2[sig] = nw_param_m2(normalW); // holds -2 w^2 (normal_weight_m2)
        if (isSpec) {
            roughA = wrcp_(lerpf(0.01f, 1.0f, sat(rough * p.roughnessFraction)));
            roughB = -rough * roughA;
        }
        sum[sig] = {c0[sig].x, c0[sig].y, c0[sig].z};
        sumVar[sig] = var;
        wsum[sig] = 1.0f;
    };
    // a signal's result out (behind the tap loop - or behind its own round)
    auto finish_signal = [&](const int sig) {
        const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
        float inv = wrcp_(wsum[sig]);
        f3 o = mul3(sum[sig], inv);
        float ov = sumVar[sig] * inv * inv;
        if (last) {
            f3 rgb = RELAX_LINEAR_RGB ? f3{fmax2(o.x, 0.0f), fmax2(o.y, 0.0f), fmax2(o.z, 0.0f)} : ycocg_to_linear(o);
            float hitDist = h2f(ld<uint16_t>(p.hist, x, y, RBPT, sig * sb + 6));
            const PlaneRef& op = isSpec ? p.outSpec : p.outDiff;
            const PlaneRef& in = isSpec ? p.inSpec : p.inDiff;
            st<uint2>(op, x, y, 8, split ? pack_h4(unpack_h4(ld<uint2>(in, x, y, 8))) : pack_h4({rgb.x, rgb.y, rgb.z, hitDist}));
            if (SH)
                st<uint2>(isSpec ? p.outSpec1 : p.outDiff1, x, y, 8, split ? pack_h4(unpack_h4(ld<uint2>(isSpec ? p.inSpec1 : p.inDiff1, x, y, 8))) : pack_h4(mul4(sum1[sig], inv)));
        } else {
            st<uint2>(p.out, x, y, RBPT, pack_h4({o.x, o.y, o.z, ov}), sig * sb);
            if (SH)
                st<uint2>(p.out, x, y, RBPT, pack_h4(mul4(sum1[sig], inv)), sig * sb + 8);
        }
    };
    constexpr bool ROUNDS = NRD_ATROUS_SIGNAL_ROUNDS_ON && NSIG == 2;
    if constexpr (!ROUNDS) {
#pragma unroll
        for (int sig = 0; sig < NSIG; sig++)
            setup_signal(sig);
    }
    const bool roughStop = p.roughnessEdgeStopping != 0;
    // the 8 taps in row-major order as a software pipeline (like k_spatial): DEPTH taps in flight, tap k is consumed right after tap
    // k + DEPTH is issued, so the arithmetic of a tap runs under the loads of the next ones (LDS flavours: the reads of a batch)
    constexpr int TI[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, TJ[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
#ifndef NRD_ATROUS_DEPTH // taps in flight of the SH gather flavour: 2 keep it at 126 VGPRs = 4 waves per SIMD (4: 132 / 3 waves; profiles/r04_ab_atrous_mask.txt)
#define NRD_ATROUS_DEPTH 2
#endif
#ifndef NRD_ATROUS_LS_DEPTH // LDS reads in flight per batch of the SH flavours that stage their window (iterations 0-2)
#define NRD_ATROUS_LS_DEPTH 4
#endif
    constexpr int DEPTH = LS ? (SH ? NRD_ATROUS_LS_DEPTH : 8) : (SH ? NRD_ATROUS_DEPTH : 8); // 32-byte SH texels: fewer taps in flight keep the kernel within its registers
    // NRD_ATROUS_SIGNAL_ROUNDS 1 (A/B): with two signals the 8 taps are walked ONCE PER SIGNAL - set-up, taps, result of the diffuse signal
    // (guide + diffuse texel per tap; the taps' geometry weight and normal distance are kept: 16 registers), then the same for the specular
    // signal (guide + specular texel, reusing them): only ONE signal's state is live at a time
    if constexpr (ROUNDS) {
#ifndef NRD_ATROUS_ROUND_DEPTH // taps in flight inside a round (gather flavours; the staged flavours read a whole round's taps from LDS at once)
#define NRD_ATROUS_ROUND_DEPTH 2
#endif
        constexpr int DEPTHR = LS ? (SH ? 4 : 8) : (SH ? NRD_ATROUS_ROUND_DEPTH : 4);
        float geoWs[8], nD2s[8];
#pragma unroll
        for (int sig = 0; sig < NSIG; sig++) {
            const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
            setup_signal(sig);
            uint2 grawR[8];
            uint2 stexR[8][SW];
            uint16_t mrawR[8];
            bool insideR[8];
            auto issueR = [&](const int k) {
                const int i = TI[k], j = TJ[k];
                if (LS) {
                    const int q = ci + (j * T + i) * LS;
                    insideR[k] = true; // outside texels were staged as sky
                    grawR[k] = sG[q];
#pragma unroll
                    for (int w = 0; w < SW; w++)
                        stexR[k][w] = sT[q * TW + sig * SW + w];
                    mrawR[k] = FIRST ? (uint16_t)(sM[q] >> (16 * sig)) : (uint16_t)0;
                    return;
                }
                int px = x + i * stride, py = y + j * stride;
                insideR[k] = ((uint32_t)px < (uint32_t)c.W) & ((uint32_t)(py - loY) <= (uint32_t)(hiY - loY));
                int cpx = imin(imax(px, 0), c.W - 1), cpy = imin(imax(py, loY), hiY);
                grawR[k] = ld_guide(p.guide, cpx, cpy);
                if constexpr (SH) {
                    const uint4 both = ld<uint4>(p.in, cpx, cpy, RBPT, sig * sb);
                    stexR[k][0] = uint2{both.x, both.y};
                    stexR[k][SW - 1] = uint2{both.z, both.w};
                } else
                    stexR[k][0] = ld<uint2>(p.in, cpx, cpy, RBPT, sig * sb);
                mrawR[k] = FIRST ? ld<uint16_t>(p.mom, cpx, cpy, LBPT, sig * 2) : (uint16_t)0;
            };
            auto consumeR = [&](const int k) {
                const int i = TI[k], j = TJ[k];
                int px = x + i * stride, gy = y + j * stride + c.yOff;
                Guide gs = decode_guide(grawR[k], c.denoisingRange);
                if (sig == 0) { // what both signals share of a tap
                    geoWs[k] = geo_weight(pg, (float)px, (float)gy, gs.z);
                    nD2s[k] = normal_dist2(normal_codes(g.nw), gs.nw);
                }
                const bool matOk = material_class(gs.mat, matFloor[sig]) == matClass[sig];
                const bool valid = insideR[k] & !gs.sky & matOk;
                float w = (i == 0 || j == 0) ? 0.5f : 0.25f;
                w *= geoWs[k];
                w *= normal_weight_m2(nD2s[k], normalW2[sig]);
                if (isSpec) {
                    float rw = smoothstep01(1.0f - absf(fma_(gs.roughness, roughA, roughB)));
                    if (LS)
                        rw = lerpf(1.0f, rw, roughRelax);
                    w *= roughStop ? rw : 1.0f;
                }
                f4 sv = unpack_h4(stexR[k][0]);
                float vs = sv.w;
                if (FIRST)
                    vs = fmax2(fma_(-signal_luma(sv, true), signal_luma(sv, true), h2f(mrawR[k])), 0.0f);
                w *= fmax2(exp_weight(absf(signal_luma(sv, true) - c0Y[sig]) * invL[sig]), minLw[sig]);
                w = valid ? w : 0.0f;
                sum[sig] = {fma_(sv.x, w, sum[sig].x), fma_(sv.y, w, sum[sig].y), fma_(sv.z, w, sum[sig].z)};
                if (SH)
                    sum1[sig] = fma4(unpack_h4(stexR[k][SW - 1]), w, sum1[sig]);
                sumVar[sig] = fma_(vs, w * w, sumVar[sig]);
                wsum[sig] += w;
            };
            if constexpr (LS) {
#pragma unroll
                for (int t0 = 0; t0 < 8; t0 += DEPTHR) {
#pragma unroll
                    for (int k = 0; k < DEPTHR; k++)
                        issueR(t0 + k);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < DEPTHR; k++)
                        consumeR(t0 + k);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < DEPTHR; k++)
                    issueR(k);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (k + DEPTHR < 8)
                        issueR(k + DEPTHR);
                    __builtin_amdgcn_sched_barrier(0);
                    consumeR(k);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            finish_signal(sig);
        }
    } else {
        uint2 graw[8];
        uint2 stex[8][RBPT / 8];
        uint16_t mraw[8][NSIG];
        bool inside[8];
        auto issue = [&](const int k) {
            const int i = TI[k], j = TJ[k];
            if (LS) {
                const int q = ci + (j * T + i) * LS;
                inside[k] = true; // outside texels were staged as sky
                graw[k] = sG[q];
    #pragma unroll
                for (int w = 0; w < TW; w++)
                    stex[k][w] = sT[q * TW + w];
    #pragma unroll
                for (int sig = 0; sig < NSIG; sig++)
                    mraw[k][sig] = FIRST ? (uint16_t)(sM[q] >> (16 * sig)) : (uint16_t)0;
                return;
            }
            int px = x + i * stride, py = y + j * stride;
            inside[k] = ((uint32_t)px < (uint32_t)c.W) & ((uint32_t)(py - loY) <= (uint32_t)(hiY - loY));
            int cpx = imin(imax(px, 0), c.W - 1), cpy = imin(imax(py, loY), hiY);
            graw[k] = ld_guide(p.guide, cpx, cpy);
            load_texel<RBPT>(p.in, cpx, cpy, stex[k]);
    #pragma unroll
            for (int sig = 0; sig < NSIG; sig++)
                mraw[k][sig] = FIRST ? ld<uint16_t>(p.mom, cpx, cpy, LBPT, sig * 2) : (uint16_t)0;
        };
        auto consume = [&](const int k) {
            const int i = TI[k], j = TJ[k];
            int px = x + i * stride, gy = y + j * stride + c.yOff;
            Guide gs = decode_guide(graw[k], c.denoisingRange);
            float geoW = geo_weight(pg, (float)px, (float)gy, gs.z);
            const float nD2 = normal_dist2(normal_codes(g.nw), gs.nw);
    #pragma unroll
            for (int sig = 0; sig < NSIG; sig++) {
                const bool isSpec = HAS_SPEC && sig == SIG_SPEC;
                // (bitwise: one basic block - the short-circuit form compiles to an exec-mask detour per operand; the material test as class
                // equality, nrd_device.h material_class) rejected taps are selected out below
                const bool matOk = material_class(gs.mat, matFloor[sig]) == matClass[sig];
                const bool valid = inside[k] & !gs.sky & matOk;
                float w = (i == 0 || j == 0) ? 0.5f : 0.25f;
                w *= geoW;
                w *= normal_weight_m2(nD2, normalW2[sig]);
                if (isSpec) {
                    float rw = smoothstep01(1.0f - absf(fma_(gs.roughness, roughA, roughB)));
                    if (LS)
                        rw = lerpf(1.0f, rw, roughRelax);
                    w *= roughStop ? rw : 1.0f;
                }
                f4 sv = unpack_h4(stex[k][sig * SW]);
                float vs = sv.w;
                if (FIRST)
                    vs = fmax2(fma_(-signal_luma(sv, true), signal_luma(sv, true), h2f(mraw[k][sig])), 0.0f);
                w *= fmax2(exp_weight(absf(signal_luma(sv, true) - c0Y[sig]) * invL[sig]), minLw[sig]);
                // a rejected tap enters with weight 0 (its texel is a finite value of an internal plane, fetched at the clamped
                // position): one select on the weight instead of one per accumulated component
                w = valid ? w : 0.0f;
                sum[sig] = {fma_(sv.x, w, sum[sig].x), fma_(sv.y, w, sum[sig].y), fma_(sv.z, w, sum[sig].z)};
                if (SH)
                    sum1[sig] = fma4(unpack_h4(stex[k][sig * SW + (SH ? 1 : 0)]), w, sum1[sig]);
                sumVar[sig] = fma_(vs, w * w, sumVar[sig]);
                wsum[sig] += w;
            }
        };
        if constexpr (LS) { // batches of DEPTH LDS reads, then their arithmetic
    #pragma unroll
            for (int t0 = 0; t0 < 8; t0 += DEPTH) {
    #pragma unroll
                for (int k = 0; k < DEPTH; k++)
                    issue(t0 + k);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int k = 0; k < DEPTH; k++)
                    consume(t0 + k);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
    #pragma unroll
            for (int k = 0; k < DEPTH; k++)
                issue(k);
            __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k + DEPTH < 8)
                    issue(k + DEPTH);
                __builtin_amdgcn_sched_barrier(0);
                consume(k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (!ROUNDS) {
#pragma unroll
        for (int sig = 0; sig < NSIG; sig++)
            finish_signal(sig);
    }
}

dim3 grid_for(const FrameConsts& c) { return dim3((unsigned)xcd_grid_blocks(c.tilesX, c.tilesY), 1, 1); }
NRD_KERNELS_END

namespace NRD_PROJ_NS {

// GRID / BLOCK: grid_for(p.c), NRD_BLK256 for kernels with barriers (one workgroup per tile), NRD_WAVE_GRID(p.c), NRD_BLK64 for the
// barrier-free ones that take their pixel from my_pixel_w (one wave per workgroup)
#define NRD_BLK256 dim3(16, 16, 1)
#if NRD_WG64
#define NRD_BLK64 dim3(16, 4, 1)
#define NRD_WAVE_GRID(c) dim3(grid_for(c).x * 4u, 1, 1)
#else
#define NRD_BLK64 dim3(16, 16, 1)
#define NRD_WAVE_GRID(c) grid_for(c)
#endif
#define NRD_LAUNCH3G(GRID, BLK, KERNEL, ...)                                                                      \
    do {                                                                                                          \
        if (p.hasDiff && p.hasSpec)                                                                               \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__ true, true>), GRID, BLK, 0, s, p);                             \
        else if (p.hasDiff)                                                                                       \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__ true, false>), GRID, BLK, 0, s, p);                            \
        else                                                                                                      \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, true>), GRID, BLK, 0, s, p);                            \
    } while (0)
#define NRD_LAUNCH3(KERNEL, ...) NRD_LAUNCH3G(grid_for(p.c), NRD_BLK256, KERNEL, __VA_ARGS__)
#define NRD_LAUNCH3W(KERNEL, ...) NRD_LAUNCH3G(NRD_WAVE_GRID(p.c), NRD_BLK64, KERNEL, __VA_ARGS__)

// same, flags AFTER the signal pair
#define NRD_LAUNCH4G(GRID, BLK, KERNEL, ...)                                                                      \
    do {                                                                                                          \
        if (p.hasDiff && p.hasSpec)                                                                               \
            hipLaunchKernelGGL((KERNEL<true, true, __VA_ARGS__>), GRID, BLK, 0, s, p);                            \
        else if (p.hasDiff)                                                                                       \
            hipLaunchKernelGGL((KERNEL<true, false, __VA_ARGS__>), GRID, BLK, 0, s, p);                           \
        else                                                                                                      \
            hipLaunchKernelGGL((KERNEL<false, true, __VA_ARGS__>), GRID, BLK, 0, s, p);                           \
    } while (0)
#define NRD_LAUNCH4(KERNEL, ...) NRD_LAUNCH4G(grid_for(p.c), NRD_BLK256, KERNEL, __VA_ARGS__)
#define NRD_LAUNCH4W(KERNEL, ...) NRD_LAUNCH4G(NRD_WAVE_GRID(p.c), NRD_BLK64, KERNEL, __VA_ARGS__)

#if NRD_PART == 1
void launch_reblur_blur_radiance(const ReblurParams& p, hipStream_t s) { NRD_LAUNCH3(k_spatial, 1, 0, ); }
#else
void launch_reblur_classify_tiles(const ReblurParams& p, hipStream_t s) {
    hipLaunchKernelGGL(k_classify_tiles, dim3((unsigned)((p.c.tilesX + NRD_CT_TILES - 1) / NRD_CT_TILES), (unsigned)p.c.tilesY, 1), dim3(16, 16, 1), 0, s, p);
}

void launch_reblur_prepare_inputs(const ReblurParams& p, hipStream_t s) {
    // the sample's default operating point has its own kernel: both radiance signals, complementary checkerboard colours, nothing else to do
    if (p.checker && p.reconRadius == 0 && !p.prepSh1 && !p.dirOcc && !p.occlusion && p.hasDiff && p.hasSpec && (p.phaseDiff ^ p.phaseSpec) == 1) {
        hipLaunchKernelGGL(k_prepare_checker, dim3((unsigned)((p.c.tilesX + NRD_CT_TILES - 1) / NRD_CT_TILES), (unsigned)p.c.tilesY, 1), dim3(16, 16, 1), 0, s, p);
        return;
    }
    NRD_LAUNCH3(k_prepare_inputs, );
}
void launch_reblur_validation(const ReblurParams& p, hipStream_t s) { hipLaunchKernelGGL(k_validation, grid_for(p.c), dim3(16, 16, 1), 0, s, p); }

void launch_reblur_spatial(const ReblurParams& p, int variant, hipStream_t s) {
    // PrePass decodes the denoiser's input convention (MODE 0..4); Blur / PostBlur only differ by the SH texel
    int mode = p.sh ? (p.relax ? 4 : 3) : (p.relax ? 1 : ((p.occlusion && !p.prepared) ? 2 : 0)); // PrepareInputs already expanded occlusion inputs to {h, 0, 0, h}
    if (variant == 0) {
        switch (mode) {
        case 0: NRD_LAUNCH3(k_spatial, 0, 0, ); break;
        case 1: NRD_LAUNCH3(k_spatial, 0, 1, ); break;
        case 2: NRD_LAUNCH3(k_spatial, 0, 2, ); break;
        case 3: NRD_LAUNCH3(k_spatial, 0, 3, ); break;
        default: NRD_LAUNCH3(k_spatial, 0, 4, ); break;
        }
    } else if (variant == 1) {
        if (p.sh)
            NRD_LAUNCH3(k_spatial, 1, 3, );
        else
            launch_reblur_blur_radiance(p, s); // own translation unit (NRD_PART 1)
    } else {
        if (p.sh)
            NRD_LAUNCH3(k_spatial, 2, 3, );
        else
            NRD_LAUNCH3(k_spatial, 2, 0, );
    }
}

void launch_reblur_prepass_temporal_accumulation(const ReblurParams& p, hipStream_t s) { NRD_LAUNCH3(k_prepass_temporal_accumulation, ); }
void launch_reblur_temporal_accumulation(const ReblurParams& p, hipStream_t s) {
    if (p.sh) {
        if (p.relax)
            NRD_LAUNCH4W(k_temporal_accumulation, true, true);
        else
            NRD_LAUNCH4W(k_temporal_accumulation, true, false);
    } else {
        if (p.relax)
            NRD_LAUNCH4W(k_temporal_accumulation, false, true);
        else
            NRD_LAUNCH4W(k_temporal_accumulation, false, false);
    }
}
void launch_reblur_history_fix(const ReblurParams& p, hipStream_t s) {
    if (p.sh)
        NRD_LAUNCH4(k_history_fix, true);
    else
        NRD_LAUNCH4(k_history_fix, false);
}
void launch_reblur_temporal_stabilization(const ReblurParams& p, hipStream_t s) {
    if (p.sh)
        NRD_LAUNCH4(k_temporal_stabilization, true);
    else
        NRD_LAUNCH4(k_temporal_stabilization, false);
}
void launch_relax_atrous(const AtrousParams& p, hipStream_t s) {
    // iteration -> (FIRST, LDS stride): 0 -> stride 1 with the moment plane, 1 -> 2, 2 -> 4, later ones gather from global memory
    if (p.sh) {
        switch (p.it) {
        case 0: NRD_LAUNCH4(k_relax_atrous, true, true, 1); break;
        case 1: NRD_LAUNCH4(k_relax_atrous, true, false, 2); break;
        case 2: NRD_LAUNCH4(k_relax_atrous, true, false, 4); break;
        default: NRD_LAUNCH4W(k_relax_atrous, true, false, 0); break;
        }
    } else {
        switch (p.it) {
        case 0: NRD_LAUNCH4(k_relax_atrous, false, true, 1); break;
        case 1: NRD_LAUNCH4(k_relax_atrous, false, false, 2); break;
        case 2: NRD_LAUNCH4(k_relax_atrous, false, false, 4); break;
        default: NRD_LAUNCH4W(k_relax_atrous, false, false, 0); break;
        }
    }
}

#endif // NRD_PART

} // namespace NRD_PROJ_NS

} // namespace nrdhip
