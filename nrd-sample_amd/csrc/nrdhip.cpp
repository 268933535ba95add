// nrdhip.cpp - host dispatch core + C-ABI of the MI355X-native NRD backend (include/nrdhip.h).
//
// This is the nrd::Instance / NRD-Integration equivalent: it turns {denoiser ids, CommonSettings, per-denoiser settings,
// bound resource slots} into an ordered list of HIP kernel launches on the caller's stream, the way the reference's
// nrd::Integration::Denoise records compute dispatches into an nri::CommandBuffer (Source/NRDSample.cpp:440-531).
// No CPU fallback exists: every pass is a HIP kernel (nrd_reblur.hip, nrd_sigma.hip).
#include "../../include/NRDDescs.h"
#include "../../include/NRDSettings.h"
#include "../../include/nrdhip.h"
#include "nrd_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>

using namespace nrdhip;

namespace {

// every kernel launcher exists per projection flavour (nrd_kernels.h): forward to the one the frame's matrices select
#define NRD_FORWARD(fn, Params) \
    void fn(const Params& p, hipStream_t s) { NRD_PICK(p.c, fn)(p, s); }
#define NRD_FORWARD_I(fn, Params) \
    void fn(const Params& p, int v, hipStream_t s) { NRD_PICK(p.c, fn)(p, v, s); }
NRD_FORWARD(launch_reference_accumulate, ReferenceParams)
NRD_FORWARD(launch_reblur_classify_tiles, ReblurParams)
NRD_FORWARD(launch_reblur_prepare_inputs, ReblurParams)
NRD_FORWARD(launch_reblur_validation, ReblurParams)
NRD_FORWARD_I(launch_reblur_spatial, ReblurParams)
NRD_FORWARD(launch_reblur_temporal_accumulation, ReblurParams)
NRD_FORWARD(launch_reblur_prepass_temporal_accumulation, ReblurParams)
NRD_FORWARD(launch_reblur_history_fix, ReblurParams)
NRD_FORWARD(launch_reblur_temporal_stabilization, ReblurParams)
NRD_FORWARD(launch_relax_atrous, AtrousParams)
NRD_FORWARD(launch_sigma_classify_tiles, SigmaParams)
NRD_FORWARD(launch_sigma_smooth_tiles, SigmaParams)
NRD_FORWARD_I(launch_sigma_blur, SigmaParams)
NRD_FORWARD(launch_sigma_temporal_stabilization, SigmaParams)
#undef NRD_FORWARD
#undef NRD_FORWARD_I

enum class Kind { REBLUR, RELAX, SIGMA, REFERENCE };

struct Plane {
    uint8_t* p = nullptr;
    uint32_t pitch = 0;
    uint32_t fmt = 0;
    uint16_t w = 0, h = 0;
    uint32_t bpt = 0;
    const char* name = "";
    bool owned = false;
    PlaneRef ref() const { return PlaneRef{p, pitch, w, h}; }
};

struct PoolPlane {
    const char* name;
    nrd::Format fmt;
    uint32_t bpt;
    uint16_t downsample;
};

uint32_t format_bytes(uint32_t f) {
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

struct Dispatch {
    const char* name;
    const char* kernel;
    uint16_t halo;
    float bpp;
    std::vector<uint32_t> written, read;
    std::function<void(hipStream_t)> launch;
    // row tiling (nrdhip_dispatch_info::read_rows): planes of `read` this dispatch fetches at the thread's own pixel only / previous-frame
    // state it fetches at motion-displaced positions / planes it reaches less far into than `halo` (e.g. a 5x5 window inside a pass whose
    // taps reach 30 rows). Everything else reports `halo`.
    std::vector<uint32_t> own, reprojected;
    std::vector<std::pair<uint32_t, uint16_t>> reach;
    bool allRows = false; // NRDHIP_DISPATCH_ALL_ROWS
    std::vector<std::pair<uint32_t, uint32_t>> prefix; // nrdhip_dispatch_info::written_prefix: {written tap-texel plane, the guide plane its texels start with}
};

// the ClassifyTiles passes run on every row the instance stores (owned + halo rows of a band): pointwise over external inputs whose
// halo rows the tiler has refreshed, so the guide and the tile planes are complete on every rank without an exchange
template <typename Params>
Params on_all_rows(Params q, int resH) {
    q.c.ownY0 = std::max(0, -q.c.yOff);
    q.c.ownY1 = std::max(q.c.ownY0, std::min(resH, q.c.H - q.c.yOff));
    q.c.tileY0 = q.c.ownY0 / 16;
    q.c.tilesY = (q.c.ownY1 + 15) / 16 - q.c.tileY0;
    q.c.tileTable = nullptr; // (the table of the owned rows' grid does not describe this one; the ClassifyTiles passes launch plain 2-D grids)
    q.c.tileFlags = nullptr;
    q.c.tilesPerXcd = 0;
    return q;
}

struct DenoiserState {
    uint32_t identifier = 0;
    nrd::Denoiser denoiser = nrd::Denoiser::MAX_NUM;
    Kind kind = Kind::REFERENCE;
    bool hasDiff = false, hasSpec = false, translucency = false, occlusion = false, sh = false;
    bool dirOcc = false; // REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION: one {direction * h, h} texel in / out, filtered as SH0 = {h,0,0,h} + SH1
    int nsig = 0;
    uint32_t permBase = 0, permEnd = 0, transBase = 0;
    uint32_t frameCounter = 0, framesSinceReset = 0;
    bool historyValid = false;
    nrd::ReblurSettings reblur;
    nrd::RelaxSettings relax;
    nrd::SigmaSettings sigma;
    nrd::ReferenceSettings reference;
    std::vector<Dispatch> dispatches;
    // ClassifyTiles' per-tile flags in launch order (FrameConsts::tileFlags), one device buffer per grid shape this denoiser ran on
    std::map<std::pair<int, int>, uint8_t*> tileFlags;
    // roughness-only terms of the specular kernel set-up by roughness code (ReblurParams::roughLut): 16 KB, REBLUR / RELAX denoisers with a
    // specular signal, allocated with the instance, rewritten by every ClassifyTiles launch
    float* roughLut = nullptr;
};

inline uint32_t enc_perm(uint32_t i) { return i; }
inline uint32_t enc_trans(uint32_t i) { return (1u << 16) | i; }
inline uint32_t enc_slot(nrd::ResourceType t) { return (2u << 16) | (uint32_t)t; }

} // namespace

struct nrdhip_instance {
    int resW = 0, resH = 0, frameH = 0, yOff = 0, ownY0 = 0, ownRows = 0;
    int histY0 = 0, histRows = 0; // nrdhip_set_history_rows (0 rows = every stored row)
    int device = -1; // HIP device ordinal the pools live on and the kernels run on (-1: whatever is current at each call)
    uint32_t flags = 0;
    nrd::CommonSettings common;
    bool commonSet = false;
    std::vector<DenoiserState> denoisers;
    std::vector<Plane> perm, trans;
    uint8_t* transArena = nullptr; // library-owned pools: the one allocation all transient planes alias into
    size_t transArenaBytes = 0;
    Plane slots[(size_t)nrd::ResourceType::MAX_NUM];
    std::string error;
    // NRDHIP_FLAG_GRAPH: one executable graph per identifier list a frame is submitted with (the sample calls Denoise three times a
    // frame - shadow, opaque, reference - each with its own list: Source/NRDSample.cpp:4082, :4126, :4224)
    // Per list a RING of executable graphs, each with the event recorded behind its last launch: an executable graph is only patched
    // (hipGraphExecUpdate) or destroyed after the launch that used it last has finished on the device - HIP does not document that an
    // in-flight launch is immune to an update of its executable graph (CUDA does), and frames are submitted back to back without a sync.
    struct GraphSlot {
        static constexpr int RING = 3;
        std::vector<uint32_t> key;
        hipGraphExec_t exec[RING] = {nullptr, nullptr, nullptr};
        hipEvent_t done[RING] = {nullptr, nullptr, nullptr};
        uint32_t next = 0;
    };
    std::vector<GraphSlot> graphExecs;
    uint32_t graphStats[3] = {0, 0, 0}; // replayed frames, instantiations, direct-launch fallbacks
    // tile order tables of the launch grids this instance has used (FrameConsts::tileTable), keyed by {tilesX, tilesY}: device memory,
    // built once per shape (a full frame has one; a row tiler adds one per strip shape), freed with the instance
    struct TileTable {
        uint32_t* dev = nullptr; // [blocks] launch index -> tile, then [tilesX * tilesY] tile -> forward launch index (the inverse)
        uint32_t blocks = 0;
        std::vector<uint32_t> host; // stays alive: the upload is an asynchronous copy on the launch stream
    };
    hipStream_t tableStream = nullptr; // the stream the current dispatch lists are launched on (uploads / clears of the tables above)
    std::map<std::pair<int, int>, TileTable> tileTables;
    bool capturing = false; // inside nrdhip_denoise's stream capture: no allocation / copy may be issued (a missing table stays missing)
};

namespace {

std::string g_createError;

DenoiserState* find(nrdhip_instance& I, uint32_t id) {
    for (auto& d : I.denoisers)
        if (d.identifier == id)
            return &d;
    return nullptr;
}

bool classify(nrd::Denoiser dn, DenoiserState& d) {
    using D = nrd::Denoiser;
    d.denoiser = dn;
    switch (dn) {
        case D::REBLUR_DIFFUSE: d.kind = Kind::REBLUR; d.hasDiff = true; break;
        case D::REBLUR_SPECULAR: d.kind = Kind::REBLUR; d.hasSpec = true; break;
        case D::REBLUR_DIFFUSE_SPECULAR: d.kind = Kind::REBLUR; d.hasDiff = d.hasSpec = true; break;
        case D::REBLUR_DIFFUSE_OCCLUSION: d.kind = Kind::REBLUR; d.hasDiff = d.occlusion = true; break;
        case D::REBLUR_SPECULAR_OCCLUSION: d.kind = Kind::REBLUR; d.hasSpec = d.occlusion = true; break;
        case D::REBLUR_DIFFUSE_SPECULAR_OCCLUSION: d.kind = Kind::REBLUR; d.hasDiff = d.hasSpec = d.occlusion = true; break;
        case D::REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION: d.kind = Kind::REBLUR; d.hasDiff = d.sh = d.dirOcc = true; break;
        case D::REBLUR_DIFFUSE_SH: d.kind = Kind::REBLUR; d.hasDiff = d.sh = true; break;
        case D::REBLUR_SPECULAR_SH: d.kind = Kind::REBLUR; d.hasSpec = d.sh = true; break;
        case D::REBLUR_DIFFUSE_SPECULAR_SH: d.kind = Kind::REBLUR; d.hasDiff = d.hasSpec = d.sh = true; break;
        case D::RELAX_DIFFUSE_SH: d.kind = Kind::RELAX; d.hasDiff = d.sh = true; break;
        case D::RELAX_SPECULAR_SH: d.kind = Kind::RELAX; d.hasSpec = d.sh = true; break;
        case D::RELAX_DIFFUSE_SPECULAR_SH: d.kind = Kind::RELAX; d.hasDiff = d.hasSpec = d.sh = true; break;
        case D::RELAX_DIFFUSE: d.kind = Kind::RELAX; d.hasDiff = true; break;
        case D::RELAX_SPECULAR: d.kind = Kind::RELAX; d.hasSpec = true; break;
        case D::RELAX_DIFFUSE_SPECULAR: d.kind = Kind::RELAX; d.hasDiff = d.hasSpec = true; break;
        case D::SIGMA_SHADOW: d.kind = Kind::SIGMA; break;
        case D::SIGMA_SHADOW_TRANSLUCENCY: d.kind = Kind::SIGMA; d.translucency = true; break;
        case D::REFERENCE: d.kind = Kind::REFERENCE; break;
        default: return false;
    }
    d.nsig = (d.hasDiff ? 1 : 0) + (d.hasSpec ? 1 : 0);
    return true;
}

// ---- pool descriptions (indices must match the enums used by the builders below) ----------------------------------
namespace rb { // REBLUR
enum Perm { GUIDE_A, GUIDE_B, DATA1_A, DATA1_B, HIST, FAST_A, FAST_B, STAB_A, STAB_B };
enum Trans { TILES, TMP1, TMP2, DATA1_TMP, DATA2, HITTRACK, PREP_D, PREP_S, PREP_D1, PREP_S1, AT_A, AT_B, // PREP_*: PrepareInputs outputs; AT_*: RELAX only
             // REBLUR only: tap texels of Blur / PostBlur (nrd_device.h), _A HistoryFix -> Blur, _B Blur -> PostBlur, one plane per signal
             TAP_D_A = AT_A, TAP_S_A, TAP_D_B, TAP_S_B };
} // namespace rb
namespace sg { // SIGMA
enum Perm { GUIDE_A, GUIDE_B, HIST_A, HIST_B };
enum Trans { TILES, TILES_SMOOTH, SHADOW1, PEN1, SHADOW2 };
} // namespace sg

// REBLUR radiance flavours run Blur / PostBlur on tap texels (guide + signal in one 16-byte texel, nrd_device.h)
bool tap_texels(const DenoiserState& d) { return d.kind == Kind::REBLUR && !d.sh; } // (OCCLUSION signals travel as {h, 0, 0, h} internally: same kernels)
// ... and their PrePass and TemporalAccumulation as ONE dispatch, unless the instance was created with NRDHIP_FLAG_SEPARATE_PASSES
bool fused_prepass(const nrdhip_instance& I, const DenoiserState& d) {
    return d.kind == Kind::REBLUR && !d.sh && !d.occlusion && !(I.flags & NRDHIP_FLAG_SEPARATE_PASSES);
}

void describe(DenoiserState& d, std::vector<PoolPlane>& perm, std::vector<PoolPlane>& trans) {
    using F = nrd::Format;
    if (d.kind == Kind::REBLUR) {
        F fmtRad = d.nsig == 2 ? F::RGBA32_UINT : F::RGBA16_SFLOAT;
        F fmtLum = d.nsig == 2 ? F::RG16_SFLOAT : F::R16_SFLOAT;
        uint32_t bRad = 8u * d.nsig * (d.sh ? 2u : 1u), bLum = 2u * d.nsig; // SH mode: SH0 + SH1 texels per signal
        perm.push_back({"REBLUR::Guide_A", F::RG32_UINT, GUIDE_BYTES, 1});
        perm.push_back({"REBLUR::Guide_B", F::RG32_UINT, GUIDE_BYTES, 1});
        perm.push_back({"REBLUR::Data1_A", F::R16_UINT, 2, 1});
        perm.push_back({"REBLUR::Data1_B", F::R16_UINT, 2, 1});
        perm.push_back({"REBLUR::History", fmtRad, bRad, 1});
        perm.push_back({"REBLUR::FastHistory_A", fmtLum, bLum, 1});
        perm.push_back({"REBLUR::FastHistory_B", fmtLum, bLum, 1});
        perm.push_back({"REBLUR::StabilizedLuma_A", fmtLum, bLum, 1});
        perm.push_back({"REBLUR::StabilizedLuma_B", fmtLum, bLum, 1});
        trans.push_back({"REBLUR::Tiles", F::R8_UINT, 1, 16});
        trans.push_back({"REBLUR::Tmp1", fmtRad, bRad, 1});
        trans.push_back({"REBLUR::Tmp2", fmtRad, bRad, 1});
        trans.push_back({"REBLUR::Data1_Tmp", F::R16_UINT, 2, 1});
        trans.push_back({"REBLUR::Data2", F::R32_UINT, 4, 1});
        trans.push_back({"REBLUR::SpecHitDistForTracking", F::R16_SFLOAT, 2, 1});
        // PrepareInputs outputs (checkerboard resolve / hit distance reconstruction): dense RGBA16F copies of the noisy
        // inputs; the SH1 copies are only full-size in SH mode
        trans.push_back({"REBLUR::Prepared_Diff", F::RGBA16_SFLOAT, 8, 1});
        trans.push_back({"REBLUR::Prepared_Spec", F::RGBA16_SFLOAT, 8, 1});
        trans.push_back({"REBLUR::Prepared_DiffSh1", F::RGBA16_SFLOAT, 8, (uint16_t)(d.sh ? 1 : 16)});
        trans.push_back({"REBLUR::Prepared_SpecSh1", F::RGBA16_SFLOAT, 8, (uint16_t)(d.sh ? 1 : 16)});
        // tap texels of Blur / PostBlur (radiance flavours only): _A HistoryFix -> Blur, _B Blur -> PostBlur
        const bool tap = tap_texels(d);
        trans.push_back({"REBLUR::Tap_Diff_A", F::RGBA32_UINT, 16, (uint16_t)(tap && d.hasDiff ? 1 : 16)});
        trans.push_back({"REBLUR::Tap_Spec_A", F::RGBA32_UINT, 16, (uint16_t)(tap && d.hasSpec ? 1 : 16)});
        trans.push_back({"REBLUR::Tap_Diff_B", F::RGBA32_UINT, 16, (uint16_t)(tap && d.hasDiff ? 1 : 16)});
        trans.push_back({"REBLUR::Tap_Spec_B", F::RGBA32_UINT, 16, (uint16_t)(tap && d.hasSpec ? 1 : 16)});
    } else if (d.kind == Kind::RELAX) { // same slot order as REBLUR for the shared front half; moments live in the STAB slots
        F fmtRad = d.nsig == 2 ? F::RGBA32_UINT : F::RGBA16_SFLOAT;
        F fmtLum = d.nsig == 2 ? F::RG16_SFLOAT : F::R16_SFLOAT;
        uint32_t bRad = 8u * d.nsig * (d.sh ? 2u : 1u), bLum = 2u * d.nsig; // SH mode: SH0 + SH1 texels per signal
        perm.push_back({"RELAX::Guide_A", F::RG32_UINT, GUIDE_BYTES, 1});
        perm.push_back({"RELAX::Guide_B", F::RG32_UINT, GUIDE_BYTES, 1});
        perm.push_back({"RELAX::HistoryLength_A", F::R16_UINT, 2, 1});
        perm.push_back({"RELAX::HistoryLength_B", F::R16_UINT, 2, 1});
        perm.push_back({"RELAX::History", fmtRad, bRad, 1});
        perm.push_back({"RELAX::FastHistory_A", fmtLum, bLum, 1});
        perm.push_back({"RELAX::FastHistory_B", fmtLum, bLum, 1});
        perm.push_back({"RELAX::Moments_A", fmtLum, bLum, 1});
        perm.push_back({"RELAX::Moments_B", fmtLum, bLum, 1});
        trans.push_back({"RELAX::Tiles", F::R8_UINT, 1, 16});
        trans.push_back({"RELAX::Tmp1", fmtRad, bRad, 1});
        trans.push_back({"RELAX::Tmp2", fmtRad, bRad, 1});
        trans.push_back({"RELAX::HistoryLength_Tmp", F::R16_UINT, 2, 1});
        trans.push_back({"RELAX::Data2", F::R32_UINT, 4, 1});
        trans.push_back({"RELAX::SpecHitDistForTracking", F::R16_SFLOAT, 2, 1});
        // PrepareInputs outputs (checkerboard resolve / hit distance reconstruction): dense RGBA16F copies of the noisy
        // inputs; the SH1 copies are only full-size in SH mode
        trans.push_back({"RELAX::Prepared_Diff", F::RGBA16_SFLOAT, 8, 1});
        trans.push_back({"RELAX::Prepared_Spec", F::RGBA16_SFLOAT, 8, 1});
        trans.push_back({"RELAX::Prepared_DiffSh1", F::RGBA16_SFLOAT, 8, (uint16_t)(d.sh ? 1 : 16)});
        trans.push_back({"RELAX::Prepared_SpecSh1", F::RGBA16_SFLOAT, 8, (uint16_t)(d.sh ? 1 : 16)});
        trans.push_back({"RELAX::Atrous_A", fmtRad, bRad, 1});
        trans.push_back({"RELAX::Atrous_B", fmtRad, bRad, 1});
    } else if (d.kind == Kind::SIGMA) {
        perm.push_back({"SIGMA::Guide_A", F::RG32_UINT, GUIDE_BYTES, 1});
        perm.push_back({"SIGMA::Guide_B", F::RG32_UINT, GUIDE_BYTES, 1});
        perm.push_back({"SIGMA::History_A", F::RGBA8_UNORM, 4, 1});
        perm.push_back({"SIGMA::History_B", F::RGBA8_UNORM, 4, 1});
        trans.push_back({"SIGMA::Tiles", F::R16_UINT, 2, 16});
        trans.push_back({"SIGMA::SmoothTiles", F::R16_UINT, 2, 16});
        trans.push_back({"SIGMA::Shadow1", F::RGBA16_SFLOAT, 8, 1});
        trans.push_back({"SIGMA::Penumbra1", F::R16_SFLOAT, 2, 1});
        trans.push_back({"SIGMA::Shadow2", F::RGBA16_SFLOAT, 8, 1});
    } else if (d.kind == Kind::REFERENCE) {
        perm.push_back({"REFERENCE::History", F::RGBA32_SFLOAT, 16, 1});
    }
}

// ---- CommonSettings -> FrameConsts (column-major 4x4 matrices, Source/NRDSample.cpp:3836-3839) ------------------------
bool derive_consts(const nrdhip_instance& I, FrameConsts& c, std::string& err) {
    const nrd::CommonSettings& cs = I.common;
    std::memset(&c, 0, sizeof(c));
    c.W = cs.rectSize[0];
    c.H = cs.rectSize[1];
    c.Wprev = cs.rectSizePrev[0] ? cs.rectSizePrev[0] : c.W;
    c.Hprev = cs.rectSizePrev[1] ? cs.rectSizePrev[1] : c.H;
    c.resW = I.resW;
    c.resH = I.resH;
    c.yOff = I.yOff;
    if (c.W <= 0 || c.H <= 0 || c.W > I.resW) {
        err = "rectSize invalid";
        return false;
    }
    if (cs.rectOrigin[0] != 0 || cs.rectOrigin[1] != 0) { // the sample never sets it (Source/NRDSample.cpp:3835-3876); refuse rather than denoise the wrong window
        err = "rectOrigin != 0 is not supported";
        return false;
    }
    if (I.frameH == I.resH && I.yOff == 0 && c.H > I.resH) {
        err = "rectSize exceeds resourceSize";
        return false;
    }
    c.ownY0 = I.ownY0;
    c.ownY1 = I.ownRows ? I.ownY0 + I.ownRows : I.resH;
    c.ownY0 = std::max(c.ownY0, -I.yOff);
    c.ownY1 = std::min(c.ownY1, c.H - I.yOff);
    c.ownY1 = std::min(c.ownY1, I.resH);
    if (c.ownY1 < c.ownY0)
        c.ownY1 = c.ownY0;
    c.prevY0 = I.histRows ? std::max(I.histY0, 0) : 0;
    c.prevY1 = I.histRows ? std::min(I.histY0 + I.histRows, I.resH) : I.resH;
    c.tilesX = (c.W + 15) / 16;
    c.tileY0 = c.ownY0 / 16;
    c.tilesY = (c.ownY1 + 15) / 16 - c.tileY0;
    c.invW = 1.0f / (float)c.W;
    c.invH = 1.0f / (float)c.H;
    c.invWprev = 1.0f / (float)c.Wprev;
    c.invHprev = 1.0f / (float)c.Hprev;

    // cameraJitter (pixels, Source/NRDSample.cpp:3843-3846): the G-buffer of pixel (x, y) was rendered through uv + jitter / rect
    // (Shaders/Composition.cs.hlsl:77 "pixelUv + gJitter") while the matrices are un-jittered - fold the constant uv offset into
    // the projection's x/y shear terms so that reconstruct / project / the tap Jacobian all see the jittered pixel grid
    // Orthographic matrices (the sample's "Ortho" camera, Source/NRDSample.cpp:1214, :1971: clip.w == 1) put the constant uv offset
    // into m12 / m13 instead; the kernels of that flavour live in nrdhip::ortho (nrd_device.h NRD_ORTHO)
    auto is_ortho = [](const float* M) { return M[11] == 0.0f && M[15] != 0.0f; };
    c.ortho = is_ortho(cs.viewToClipMatrix) ? 1 : 0;
    auto projection = [](const float* M, float* pj, float* fr, const float* jitter, float invW, float invH, bool ortho) {
        if (M[0] == 0.0f || M[5] == 0.0f)
            return false; // degenerate
        if (ortho) {
            float w = M[15];
            float m0 = M[0] / w, m5 = M[5] / w;
            float m12 = M[12] / w - 2.0f * jitter[0] * invW, m13 = M[13] / w + 2.0f * jitter[1] * invH;
            pj[0] = m0;
            pj[1] = m5;
            pj[2] = m12;
            pj[3] = m13;
            pj[4] = 1.0f;
            fr[2] = 2.0f / m0;
            fr[0] = (-1.0f - m12) / m0;
            fr[3] = -2.0f / m5;
            fr[1] = (1.0f - m13) / m5;
            return true;
        }
        float s = M[11];
        if (s == 0.0f)
            return false;
        s = s > 0.0f ? 1.0f : -1.0f;
        float m8 = M[8] - 2.0f * s * jitter[0] * invW, m9 = M[9] + 2.0f * s * jitter[1] * invH;
        pj[0] = M[0];
        pj[1] = M[5];
        pj[2] = m8;
        pj[3] = m9;
        pj[4] = s;
        fr[2] = 2.0f * s / M[0];
        fr[0] = (-s - m8) / M[0];
        fr[3] = -2.0f * s / M[5];
        fr[1] = (s - m9) / M[5];
        return true;
    };
    // a projection-mode switch between frames comes with an accumulation restart (Source/NRDSample.cpp:2142): the previous
    // projection then only has to be well-formed, so the current one stands in for it
    const bool prevSameMode = (is_ortho(cs.viewToClipMatrixPrev) ? 1 : 0) == c.ortho;
    const float* Mprev = prevSameMode ? cs.viewToClipMatrixPrev : cs.viewToClipMatrix;
    if (!projection(cs.viewToClipMatrix, c.pj, c.fr, cs.cameraJitter, c.invW, c.invH, c.ortho != 0) ||
        !projection(Mprev, c.pjPrev, c.frPrev, cs.cameraJitterPrev, c.invWprev, c.invHprev, c.ortho != 0)) {
        err = "viewToClipMatrix is neither a perspective nor an orthographic projection";
        return false;
    }
    for (int k = 0; k < 2; k++) {
        float* pv = k ? c.pvPrev : c.pv;
        const float* fr = k ? c.frPrev : c.fr;
        float iw = k ? c.invWprev : c.invW, ih = k ? c.invHprev : c.invH;
        pv[2] = fr[2] * iw;
        pv[3] = fr[3] * ih;
        pv[0] = fr[0] + 0.5f * pv[2];
        pv[1] = fr[1] + 0.5f * pv[3];
    }
    auto camera = [](const float* M, float* R, float* Rt, float* pos) {
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) {
                R[r * 3 + k] = M[k * 4 + r];
                Rt[k * 3 + r] = M[k * 4 + r];
            }
        for (int i = 0; i < 3; i++)
            pos[i] = -(R[0 * 3 + i] * M[12] + R[1 * 3 + i] * M[13] + R[2 * 3 + i] * M[14]);
    };
    float pos[3], posPrev[3];
    camera(cs.worldToViewMatrix, c.w2v, c.v2w, pos);
    camera(cs.worldToViewMatrixPrev, c.w2vPrev, c.v2wPrev, posPrev);
    for (int i = 0; i < 3; i++)
        c.camDelta[i] = posPrev[i] - pos[i];
    c.unproject = 1.0f / (0.5f * (float)c.H * std::fabs(c.pj[1]));
    c.minRectDimMulUnproject = (float)std::min(c.W, c.H) * c.unproject;
    c.jcx = 0.5f * (float)c.W * c.pj[0] * c.unproject * c.pj[4]; // (pj[4] = +-1: 1 / pj[4] == pj[4]; nrd_device.h kernel_basis_px)
    c.jcy = -0.5f * (float)c.H * c.pj[1] * c.unproject * c.pj[4];
    c.denoisingRange = cs.denoisingRange;
    c.disocclusionThreshold = cs.disocclusionThreshold;
    c.disoccAlt = cs.disocclusionThresholdAlternate;
    c.mixAvail = cs.isDisocclusionThresholdMixAvailable ? 1 : 0;
    c.splitScreen = cs.splitScreen;
    for (int i = 0; i < 3; i++)
        c.mvScale[i] = cs.motionVectorScale[i];
    c.viewZScale = cs.viewZScale;
    c.frameIndex = cs.frameIndex;
    c.strandMat = (cs.strandMaterialID >= 0.0f && cs.strandMaterialID <= 3.0f) ? (uint32_t)cs.strandMaterialID : 0xffffffffu; // 2-bit material IDs
    c.strandThickness = cs.strandThickness;
    c.camAttachMat = (cs.cameraAttachedReflectionMaterialID >= 0.0f && cs.cameraAttachedReflectionMaterialID <= 3.0f) ? (uint32_t)cs.cameraAttachedReflectionMaterialID : 0xffffffffu;
    c.mvWorld = cs.isMotionVectorInWorldSpace ? 1 : 0;
    c.confAvail = cs.isHistoryConfidenceAvailable ? 1 : 0;
    for (int k = 0; k < 64; k++) {
        double a = 6.283185307179586 * (double)k / 64.0;
        c.rot[k][0] = (float)std::cos(a);
        c.rot[k][1] = (float)std::sin(a);
    }
    return true;
}

// ---- dispatch lists ------------------------------------------------------------------------------------------------
// Direction in which a dispatch walks the tiles (FrameConsts::reverse, nrd_device.h xcd_tile_kj). A reader that walks the frame AGAINST
// its writer starts in what the 256 MiB Infinity Cache still holds of a 133-266 MB plane; one that follows the writer chases the
// eviction front. Measured per pass at 4K (profiles/r03_ab_traversal_direction.txt): reversing HistoryFix alone (it reads what
// TemporalAccumulation wrote, and Blur then reads HistoryFix's tap texels against ITS direction) takes HistoryFix -8 % and Blur -6 %;
// reversing any other REBLUR pass costs that pass 2-3 % or buys nothing, so the direction is a per-pass property, not an alternation.
#ifndef NRD_REVERSE_HISTORY_FIX
#define NRD_REVERSE_HISTORY_FIX 1
#endif
#ifndef NRD_ATROUS_ALTERNATE // 1: odd RELAX A-trous iterations walk the tiles back to front when the planes outgrow the Infinity Cache
#define NRD_ATROUS_ALTERNATE 1
#endif

template <typename Params>
Params directed(Params q, bool reverse) {
    q.c.reverse = reverse ? 1 : 0;
    return q;
}

// a stream capture in progress on `st` - the library's own (NRDHIP_FLAG_GRAPH) or the caller's: no allocation, no copy may be issued
bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &status) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return status != hipStreamCaptureStatusNone;
}

// ClassifyTiles' flags in launch order for the passes behind it (FrameConsts::tileFlags): usable when those passes launch over the grid
// ClassifyTiles runs on (one instance holding the whole frame; a band of a row tiler classifies its halo rows too) and that grid has a table
void attach_tile_flags(nrdhip_instance& I, DenoiserState& d, ReblurParams& p) {
    p.c.tileFlags = nullptr;
    p.tileFlagsOut = nullptr;
    p.tileInv = nullptr;
    const ReblurParams q = on_all_rows(p, I.resH);
    if (!p.c.tileTable || q.c.tileY0 != p.c.tileY0 || q.c.tilesY != p.c.tilesY)
        return;
    auto t = I.tileTables.find({p.c.tilesX, p.c.tilesY});
    if (t == I.tileTables.end() || !t->second.dev)
        return;
    uint8_t* f = nullptr;
    auto it = d.tileFlags.find({p.c.tilesX, p.c.tilesY});
    if (it != d.tileFlags.end())
        f = it->second;
    else if (!I.capturing && !stream_is_capturing(I.tableStream)) {
        const size_t bytes = (size_t)t->second.blocks + 8; // (the scalar read fetches the aligned dword around a byte)
        if (hipMalloc((void**)&f, bytes) != hipSuccess || hipMemset(f, 0, bytes) != hipSuccess) { // (blocking, once per shape: complete for every stream)
            (void)hipGetLastError();
            if (f)
                (void)hipFree(f);
            f = nullptr;
        }
        d.tileFlags[{p.c.tilesX, p.c.tilesY}] = f;
    }
    if (!f)
        return;
    p.c.tileFlags = f;
    p.tileFlagsOut = f;
    p.tileInv = t->second.dev + t->second.blocks;
}

void build_reference(nrdhip_instance& I, DenoiserState& d, const FrameConsts& c) {
    using RT = nrd::ResourceType;
    ReferenceParams p;
    p.c = c;
    bool restart = !c.historyOk;
    uint32_t n = std::min(d.framesSinceReset, d.reference.maxAccumulatedFrameNum);
    p.weight = restart ? 1.0f : 1.0f / (1.0f + (float)n);
    p.restart = restart ? 1 : 0;
    p.in = I.slots[(size_t)RT::IN_SIGNAL].ref();
    p.out = I.slots[(size_t)RT::OUT_SIGNAL].ref();
    p.hist = I.perm[d.permBase].ref();
    Dispatch x{"REFERENCE::TemporalAccumulation", "nrd_reference_accumulate", 0, 8 + 16 + 16 + 8, {}, {}, nullptr};
    x.read = {enc_slot(RT::IN_SIGNAL), enc_perm(d.permBase)};
    x.written = {enc_perm(d.permBase), enc_slot(RT::OUT_SIGNAL)};
    x.launch = [p](hipStream_t s) { launch_reference_accumulate(p, s); };
    d.dispatches.push_back(x);
}

nrd::ResourceType in_slot(const DenoiserState& d, bool spec) {
    using RT = nrd::ResourceType;
    if (d.dirOcc)
        return RT::IN_DIFF_DIRECTION_HITDIST;
    if (d.sh)
        return spec ? RT::IN_SPEC_SH0 : RT::IN_DIFF_SH0;
    return d.occlusion ? (spec ? RT::IN_SPEC_HITDIST : RT::IN_DIFF_HITDIST) : (spec ? RT::IN_SPEC_RADIANCE_HITDIST : RT::IN_DIFF_RADIANCE_HITDIST);
}
nrd::ResourceType out_slot(const DenoiserState& d, bool spec) {
    using RT = nrd::ResourceType;
    if (d.dirOcc)
        return RT::OUT_DIFF_DIRECTION_HITDIST;
    if (d.sh)
        return spec ? RT::OUT_SPEC_SH0 : RT::OUT_DIFF_SH0;
    return d.occlusion ? (spec ? RT::OUT_SPEC_HITDIST : RT::OUT_DIFF_HITDIST) : (spec ? RT::OUT_SPEC_RADIANCE_HITDIST : RT::OUT_DIFF_RADIANCE_HITDIST);
}
// PrepareInputs (nrd_reblur.hip k_prepare_inputs) is recorded when the inputs are checkerboarded or hit distances must be
// reconstructed (Source/NRDSample.cpp:545-548: the sample's default RESOLUTION_HALF tracing sets CheckerboardMode::WHITE)
struct PrepareMode {
    bool any, checker;
    int phase[2]; // per signal (0 diffuse, 1 specular): Sequence::CheckerBoard value carrying it, 2 = every pixel
    int radius;
    bool sh1; // the SH1 texels are (re)written too: checkerboarded SH inputs, or DIRECTIONAL_OCCLUSION (its single
              // {direction * h, h} input texel is always split into SH0 = {h,0,0,h} and SH1 = {direction * h, 0} there)
};
PrepareMode prepare_mode(const DenoiserState& d, const nrd::ReblurSettings& s) {
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
// the planes the PrePass gathers its signals from: the input slots, or the PrepareInputs copies
void push_prepass_inputs(const DenoiserState& d, const PrepareMode& pm, uint32_t tb, std::vector<uint32_t>& list) {
    using RT = nrd::ResourceType;
    for (int spec = 0; spec < 2; spec++) {
        if (spec ? !d.hasSpec : !d.hasDiff)
            continue;
        list.push_back(pm.any ? enc_trans(tb + rb::PREP_D + spec) : enc_slot(in_slot(d, spec != 0)));
        if (d.sh)
            list.push_back(pm.sh1 ? enc_trans(tb + rb::PREP_D1 + spec) : enc_slot(spec ? RT::IN_SPEC_SH1 : RT::IN_DIFF_SH1));
    }
}
// the PrepareInputs dispatch itself (shared by REBLUR and RELAX)
void push_prepare_dispatch(DenoiserState& d, const PrepareMode& pm, const ReblurParams& p, const char* name, uint32_t guide, uint32_t tb) {
    using RT = nrd::ResourceType;
    float n = (float)d.nsig;
    float inB = (d.occlusion ? 2.0f : 8.0f) * (pm.checker ? 0.5f : 1.0f);
    Dispatch x{name, "nrd_reblur_prepare_inputs", (uint16_t)pm.radius, (float)GUIDE_BYTES + n * (inB + 8.0f) + (pm.sh1 ? n * ((d.dirOcc ? 0.0f : 4.0f) + 8.0f) : 0.0f), {}, {}, nullptr};
    x.read = {guide};
    for (int spec = 0; spec < 2; spec++) {
        if (spec ? !d.hasSpec : !d.hasDiff)
            continue;
        x.read.push_back(enc_slot(in_slot(d, spec != 0)));
        x.written.push_back(enc_trans(tb + rb::PREP_D + spec));
        if (pm.sh1) {
            if (!d.dirOcc)
                x.read.push_back(enc_slot(spec ? RT::IN_SPEC_SH1 : RT::IN_DIFF_SH1));
            x.written.push_back(enc_trans(tb + rb::PREP_D1 + spec));
        }
    }
    x.launch = [p](hipStream_t st) { launch_reblur_prepare_inputs(p, st); };
    d.dispatches.push_back(x);
}

// validation overlay dispatch (REBLUR and RELAX), recorded only when asked for and OUT_VALIDATION is bound
void push_validation_dispatch(nrdhip_instance& I, DenoiserState& d, const ReblurParams& p, const char* name, uint32_t guide, uint32_t data1, uint32_t data2) {
    if (!I.common.enableValidation || !I.slots[(size_t)nrd::ResourceType::OUT_VALIDATION].p)
        return;
    Dispatch x{name, "nrd_reblur_validation", 0, (float)GUIDE_BYTES + 2.0f + 4.0f + 4.0f, {}, {}, nullptr};
    x.read = {guide, data1, data2};
    x.written = {enc_slot(nrd::ResourceType::OUT_VALIDATION)};
    x.launch = [p](hipStream_t st) { launch_reblur_validation(p, st); };
    d.dispatches.push_back(x);
}

// appends the signal slots (SH0 [+ SH1]) of the active signals to a dispatch's read / written list
void push_signal_slots(const DenoiserState& d, std::vector<uint32_t>& list, bool outputs) {
    using RT = nrd::ResourceType;
    for (int spec = 0; spec < 2; spec++) {
        if (spec ? !d.hasSpec : !d.hasDiff)
            continue;
        list.push_back(enc_slot(outputs ? out_slot(d, spec != 0) : in_slot(d, spec != 0)));
        if (d.sh && !d.dirOcc)
            list.push_back(enc_slot(outputs ? (spec ? RT::OUT_SPEC_SH1 : RT::OUT_DIFF_SH1) : (spec ? RT::IN_SPEC_SH1 : RT::IN_DIFF_SH1)));
    }
}

// the tap planes of the signals present, diffuse first (base = rb::TAP_D_A or rb::TAP_D_B)
void push_tap_planes(const DenoiserState& d, uint32_t tb, int base, std::vector<uint32_t>& list) {
    if (d.hasDiff)
        list.push_back(enc_trans(tb + base));
    if (d.hasSpec)
        list.push_back(enc_trans(tb + base + 1));
}

// parameter block of the REBLUR kernels; RELAX reuses them for its front half (ClassifyTiles, PrePass, TA, HistoryFix)
ReblurParams make_reblur_params(nrdhip_instance& I, DenoiserState& d, const FrameConsts& c, const nrd::ReblurSettings& s) {
    using RT = nrd::ResourceType;
    int cur = (int)(d.frameCounter & 1);
    uint32_t pb = d.permBase, tb = d.transBase;
    auto PP = [&](int i) { return I.perm[pb + i].ref(); };
    auto TP = [&](int i) { return I.trans[tb + i].ref(); };
    auto SP = [&](RT t) { return I.slots[(size_t)t].ref(); };

    ReblurParams p;
    std::memset(&p, 0, sizeof(p));
    p.c = c;
    p.hp[0] = s.hitDistanceParameters.A;
    p.hp[1] = s.hitDistanceParameters.B;
    p.hp[2] = s.hitDistanceParameters.C;
    p.hp[3] = s.hitDistanceParameters.D;
    p.hitFactorDiff = reblur_hitdist_factor(p.hp, 1.0f);
    p.roughLut = d.roughLut;
    p.planeDistanceSensitivity = s.planeDistanceSensitivity;
    p.lobeAngleFraction = s.lobeAngleFraction;
    p.roughnessFraction = s.roughnessFraction;
    p.minHitDistanceWeight = s.minHitDistanceWeight;
    p.minBlurRadius = s.minBlurRadius;
    p.maxBlurRadius = s.maxBlurRadius;
    p.diffusePrepassBlurRadius = s.diffusePrepassBlurRadius;
    p.specularPrepassBlurRadius = s.specularPrepassBlurRadius;
    p.fastHistoryClampingSigmaScale = s.fastHistoryClampingSigmaScale;
    p.antilagSigmaScale = s.antilagSettings.luminanceSigmaScale;
    p.antilagSensitivity = s.antilagSettings.luminanceSensitivity;
    p.responsiveRoughnessThreshold = s.responsiveAccumulationSettings.roughnessThreshold;
    p.responsiveMinAccum = (float)s.responsiveAccumulationSettings.minAccumulatedFrameNum;
    p.maxA = (float)std::min<uint32_t>(s.maxAccumulatedFrameNum, 63);
    p.maxFastA = (float)std::min<uint32_t>(s.maxFastAccumulatedFrameNum, 63);
    p.maxStab = (float)std::min<uint32_t>(s.maxStabilizedFrameNum, 63);
    p.prepassTrackOnly = s.usePrepassOnlyForSpecularMotionEstimation ? 1 : 0;
    p.returnHistLen = (d.occlusion && s.returnHistoryLengthInsteadOfOcclusion) ? 1 : 0;
    p.invMaxA = 1.0f / std::max(p.maxA, 1.0f);
    p.historyFixFrameNum = (int)s.historyFixFrameNum;
    p.historyFixStride = (int)s.historyFixBasePixelStride;
    p.reachPre = (int)(std::max(s.diffusePrepassBlurRadius, s.specularPrepassBlurRadius) * 1.1f) + 3;
    p.reachBlur = (int)((s.maxBlurRadius + s.minBlurRadius) * 1.1f) + 3;
    p.reachPost = (int)((s.maxBlurRadius + s.minBlurRadius) * 2.2f) + 3;
    static const float disk[8][3] = NRD_POISSON8_TABLE;
    rotate_taps(c.rot, c.frameIndex, 1u, disk, p.tapsPre);  // salt = VARIANT + 1 (nrd_reblur.hip k_spatial)
    rotate_taps(c.rot, c.frameIndex, 3u, disk, p.tapsPost);
    p.minMatDiff = s.minMaterialForDiffuse;
    p.minMatSpec = s.minMaterialForSpecular;
    p.clampEnabled = s.maxFastAccumulatedFrameNum < s.maxAccumulatedFrameNum ? 1 : 0;
    p.antiFirefly = s.enableAntiFirefly ? 1 : 0;
    p.fireflyScale = s.fireflySuppressorMinRelativeScale;
    p.hasDiff = d.hasDiff;
    p.hasSpec = d.hasSpec;
    p.inZ = SP(RT::IN_VIEWZ);
    p.inNR = SP(RT::IN_NORMAL_ROUGHNESS);
    p.inMV = SP(RT::IN_MV);
    p.inMix = SP(RT::IN_DISOCCLUSION_THRESHOLD_MIX);
    p.inDiff = SP(in_slot(d, false));
    p.inSpec = SP(in_slot(d, true));
    p.outValidation = SP(RT::OUT_VALIDATION);
    p.inDiff1 = SP(RT::IN_DIFF_SH1);
    p.inSpec1 = SP(RT::IN_SPEC_SH1);
    // PrepareInputs reads the slots ("raw") and hands dense copies to the PrePass
    PrepareMode pm = prepare_mode(d, s);
    p.rawDiff = p.inDiff;
    p.rawSpec = p.inSpec;
    p.rawDiff1 = p.inDiff1;
    p.rawSpec1 = p.inSpec1;
    p.prepared = pm.any ? 1 : 0;
    p.checker = pm.checker ? 1 : 0;
    p.phaseDiff = pm.phase[0];
    p.phaseSpec = pm.phase[1];
    p.reconRadius = pm.radius;
    p.prepSh1 = pm.sh1 ? 1 : 0;
    p.dirOcc = d.dirOcc ? 1 : 0;
    p.dirInSnorm = (d.dirOcc && I.slots[(size_t)nrd::ResourceType::IN_DIFF_DIRECTION_HITDIST].fmt == (uint32_t)nrd::Format::RGBA16_SNORM) ? 1 : 0;
    p.dirOutSnorm = (d.dirOcc && I.slots[(size_t)nrd::ResourceType::OUT_DIFF_DIRECTION_HITDIST].fmt == (uint32_t)nrd::Format::RGBA16_SNORM) ? 1 : 0;
    if (pm.any) {
        p.inDiff = TP(rb::PREP_D);
        p.inSpec = TP(rb::PREP_S);
        if (pm.sh1) {
            p.inDiff1 = TP(rb::PREP_D1);
            p.inSpec1 = TP(rb::PREP_S1);
        }
    }
    p.outDiff1 = SP(RT::OUT_DIFF_SH1);
    p.outSpec1 = SP(RT::OUT_SPEC_SH1);
    p.sh = d.sh ? 1 : 0;
    p.occlusion = d.occlusion ? 1 : 0;
    p.ioF16 = I.slots[(size_t)in_slot(d, !d.hasDiff)].fmt == (uint32_t)nrd::Format::R16_SFLOAT ? 1 : 0;
    p.confD = SP(RT::IN_DIFF_CONFIDENCE);
    p.confS = SP(RT::IN_SPEC_CONFIDENCE);
    p.outDiff = SP(out_slot(d, false));
    p.outSpec = SP(out_slot(d, true));
    p.guide = PP(rb::GUIDE_A + cur);
    p.guidePrev = PP(rb::GUIDE_A + (cur ^ 1));
    p.data1 = PP(rb::DATA1_A + cur);
    p.data1Prev = PP(rb::DATA1_A + (cur ^ 1));
    p.hist = PP(rb::HIST);
    p.fast = PP(rb::FAST_A + cur);
    p.fastPrev = PP(rb::FAST_A + (cur ^ 1));
    p.stab = PP(rb::STAB_A + cur);
    p.stabPrev = PP(rb::STAB_A + (cur ^ 1));
    p.tiles = TP(rb::TILES);
    p.tmp1 = TP(rb::TMP1);
    p.tmp2 = TP(rb::TMP2);
    p.data1Tmp = TP(rb::DATA1_TMP);
    p.data2 = TP(rb::DATA2);
    p.hitTrack = TP(rb::HITTRACK);
    p.tapTex = tap_texels(d) ? 1 : 0;
    if (p.tapTex)
        for (int sgl = 0; sgl < 2; sgl++) {
            p.tapA[sgl] = TP(rb::TAP_D_A + sgl);
            p.tapB[sgl] = TP(rb::TAP_D_B + sgl);
        }
    p.maxASpec = p.maxA;
    p.maxFastASpec = p.maxFastA;
    p.relax = 0;
    return p;
}

void build_reblur(nrdhip_instance& I, DenoiserState& d, const FrameConsts& c) {
    using RT = nrd::ResourceType;
    const nrd::ReblurSettings& s = d.reblur;
    int cur = (int)(d.frameCounter & 1);
    uint32_t pb = d.permBase, tb = d.transBase;
    auto P = [&](int i) { return enc_perm(pb + i); };
    auto T = [&](int i) { return enc_trans(tb + i); };
    ReblurParams p = make_reblur_params(I, d, c, s);
    attach_tile_flags(I, d, p);

    float n = (float)d.nsig;
    float nr = n * (d.sh ? 2.0f : 1.0f); // radiance texels per pixel (SH mode doubles them)
    float sp = d.hasSpec ? 2.0f : 0.0f;
    uint16_t blurHalo = (uint16_t)p.reachBlur, postHalo = (uint16_t)p.reachPost, preHalo = (uint16_t)p.reachPre;
    const float GB = (float)GUIDE_BYTES; // guide texel bytes
    const bool tap = tap_texels(d);
    {
        Dispatch x{"REBLUR::ClassifyTiles", "nrd_reblur_classify_tiles", 0, 4 + 4 + GB + 1.0f / 256.0f, {}, {}, nullptr};
        x.read = {enc_slot(RT::IN_VIEWZ), enc_slot(RT::IN_NORMAL_ROUGHNESS)};
        x.written = {P(rb::GUIDE_A + cur), T(rb::TILES)};
        x.allRows = true;
        { auto q = on_all_rows(p, I.resH); x.launch = [q](hipStream_t st) { launch_reblur_classify_tiles(q, st); }; } // (plain grid, no direction: does not take part in the alternation)
        d.dispatches.push_back(x);
    }
    const PrepareMode pm = prepare_mode(d, s);
    if (pm.any)
        push_prepare_dispatch(d, pm, p, "REBLUR::PrepareInputs", P(rb::GUIDE_A + cur), tb);
    if (fused_prepass(I, d)) {
        // PrePass + TemporalAccumulation in one launch (nrd_reblur.hip spatial_pixel<..., FUSED>): TemporalAccumulation reads the PrePass
        // result at its own pixel only, so it stays in registers - Tmp1 is neither written nor read, the guide is fetched once
        Dispatch x{"REBLUR::PrePassTemporalAccumulation", "nrd_reblur_prepass_temporal_accumulation", preHalo,
                   GB + 8 * nr + 8 + GB + 2 + 8 * nr + 2 * n + 8 * nr + 2 * n + 2 + 4 + sp, {}, {}, nullptr};
        x.read = {P(rb::GUIDE_A + cur)};
        push_prepass_inputs(d, pm, tb, x.read);
        for (uint32_t r : {P(rb::GUIDE_A + (cur ^ 1)), enc_slot(RT::IN_MV), P(rb::HIST), P(rb::FAST_A + (cur ^ 1)), P(rb::DATA1_A + (cur ^ 1))})
            x.read.push_back(r);
        if (c.mixAvail)
            x.read.push_back(enc_slot(RT::IN_DISOCCLUSION_THRESHOLD_MIX));
        x.written = {T(rb::HITTRACK), T(rb::TMP2), P(rb::FAST_A + cur), T(rb::DATA1_TMP), T(rb::DATA2)};
        x.reprojected = {P(rb::GUIDE_A + (cur ^ 1)), P(rb::HIST), P(rb::FAST_A + (cur ^ 1)), P(rb::DATA1_A + (cur ^ 1))};
        x.launch = [p](hipStream_t st) { launch_reblur_prepass_temporal_accumulation(p, st); };
        d.dispatches.push_back(x);
    } else {
        Dispatch x{"REBLUR::PrePass", "nrd_reblur_prepass", preHalo, GB + 8 * nr + 8 * nr + sp, {}, {}, nullptr};
        x.read = {P(rb::GUIDE_A + cur)};
        push_prepass_inputs(d, pm, tb, x.read);
        x.written = {T(rb::TMP1), T(rb::HITTRACK)};
        x.launch = [p](hipStream_t st) { launch_reblur_spatial(p, 0, st); };
        d.dispatches.push_back(x);
    }
    if (!fused_prepass(I, d)) {
        Dispatch x{"REBLUR::TemporalAccumulation", "nrd_reblur_temporal_accumulation", 0,
                   GB + 8 + GB + 2 + 8 * nr + 8 * nr + 2 * n + sp + 8 * nr + 2 * n + 2 + 4, {}, {}, nullptr};
        x.read = {P(rb::GUIDE_A + cur), P(rb::GUIDE_A + (cur ^ 1)), enc_slot(RT::IN_MV), T(rb::TMP1), P(rb::HIST),
                  P(rb::FAST_A + (cur ^ 1)), P(rb::DATA1_A + (cur ^ 1)), T(rb::HITTRACK)};
        if (c.mixAvail)
            x.read.push_back(enc_slot(RT::IN_DISOCCLUSION_THRESHOLD_MIX));
        x.written = {T(rb::TMP2), P(rb::FAST_A + cur), T(rb::DATA1_TMP), T(rb::DATA2)};
        x.reprojected = {P(rb::GUIDE_A + (cur ^ 1)), P(rb::HIST), P(rb::FAST_A + (cur ^ 1)), P(rb::DATA1_A + (cur ^ 1))};
        x.launch = [p](hipStream_t st) { launch_reblur_temporal_accumulation(p, st); };
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"REBLUR::HistoryFix", "nrd_reblur_history_fix", (uint16_t)(2 * s.historyFixBasePixelStride + 2),
                   GB + 2 + 8 * nr + 2 * n + (tap ? 16 * n : 8 * nr) + 2, {}, {}, nullptr};
        // (HistoryFix / TemporalStabilization also look up the Tiles flag of their OWN tile - no reach across tiles or bands, so it
        // is not part of the exchange plan's read sets)
        x.read = {P(rb::GUIDE_A + cur), T(rb::TMP2), T(rb::DATA1_TMP), P(rb::FAST_A + cur)};
        x.reach = {{P(rb::FAST_A + cur), (uint16_t)2}}; // the 5x5 clamping window; the reconstruction taps read guide, signal and speeds
        if (tap) {
            push_tap_planes(d, tb, rb::TAP_D_A, x.written);
            for (uint32_t code : x.written) // "the tap texels carry the pixel's guide texel as it is" (k_history_fix)
                x.prefix.push_back({code, P(rb::GUIDE_A + cur)});
            x.written.push_back(P(rb::DATA1_A + cur));
        } else
            x.written = {T(rb::TMP1), P(rb::DATA1_A + cur)};
        { auto q = directed(p, NRD_REVERSE_HISTORY_FIX != 0); x.launch = [q](hipStream_t st) { launch_reblur_history_fix(q, st); }; }
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"REBLUR::Blur", "nrd_reblur_blur", blurHalo, tap ? 2 + 16 * n + 16 * n : GB + 2 + 8 * nr + 8 * nr, {}, {}, nullptr};
        if (tap) { // the tap texels carry the guide: no guide plane access
            x.read = {P(rb::DATA1_A + cur)};
            push_tap_planes(d, tb, rb::TAP_D_A, x.read);
            push_tap_planes(d, tb, rb::TAP_D_B, x.written);
            for (uint32_t code : x.written) // Blur hands the guide part of its input texel on
                x.prefix.push_back({code, P(rb::GUIDE_A + cur)});
        } else {
            x.read = {P(rb::GUIDE_A + cur), P(rb::DATA1_A + cur), T(rb::TMP1)};
            x.written = {T(rb::TMP2)};
        }
        x.own = {P(rb::DATA1_A + cur)};
        x.launch = [p](hipStream_t st) { launch_reblur_spatial(p, 1, st); };
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"REBLUR::PostBlur", "nrd_reblur_post_blur", postHalo, tap ? 2 + 16 * n + 8 * nr : GB + 2 + 8 * nr + 8 * nr, {}, {}, nullptr};
        if (tap) {
            x.read = {P(rb::DATA1_A + cur)};
            push_tap_planes(d, tb, rb::TAP_D_B, x.read);
        } else
            x.read = {P(rb::GUIDE_A + cur), P(rb::DATA1_A + cur), T(rb::TMP2)};
        x.written = {P(rb::HIST)};
        x.own = {P(rb::DATA1_A + cur)};
        x.launch = [p](hipStream_t st) { launch_reblur_spatial(p, 2, st); };
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"REBLUR::TemporalStabilization", "nrd_reblur_temporal_stabilization", 2,
                   GB + 2 + 4 + 8 + 8 * nr + 2 * n + sp + 8 * nr + 2 * n, {}, {}, nullptr};
        x.read = {P(rb::GUIDE_A + cur), P(rb::DATA1_A + cur), T(rb::DATA2), enc_slot(RT::IN_MV), P(rb::HIST), P(rb::STAB_A + (cur ^ 1)), T(rb::HITTRACK)};
        x.written = {P(rb::STAB_A + cur)};
        push_signal_slots(d, x.written, true);
        push_signal_slots(d, x.read, false);
        x.own = {P(rb::DATA1_A + cur), T(rb::DATA2), T(rb::HITTRACK)}; // (guide and history: the 5x5 window, = the pass's halo)
        x.reprojected = {P(rb::STAB_A + (cur ^ 1))};
        x.launch = [p](hipStream_t st) { launch_reblur_temporal_stabilization(p, st); };
        d.dispatches.push_back(x);
    }
    push_validation_dispatch(I, d, p, "REBLUR::Validation", P(rb::GUIDE_A + cur), P(rb::DATA1_A + cur), T(rb::DATA2));
}

nrd::ReblurSettings relax_as_reblur(const nrd::RelaxSettings& r) {
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

// RELAX = shared front half (ClassifyTiles, PrePass, TemporalAccumulation, HistoryFix) + variance-guided A-trous iterations
void build_relax(nrdhip_instance& I, DenoiserState& d, const FrameConsts& c) {
    using RT = nrd::ResourceType;
    const nrd::RelaxSettings& r = d.relax;
    nrd::ReblurSettings s = relax_as_reblur(r);
    int cur = (int)(d.frameCounter & 1);
    uint32_t pb = d.permBase, tb = d.transBase;
    auto P = [&](int i) { return enc_perm(pb + i); };
    auto T = [&](int i) { return enc_trans(tb + i); };
    ReblurParams p = make_reblur_params(I, d, c, s);
    attach_tile_flags(I, d, p);
    p.relax = 1;
    p.maxASpec = (float)std::min<uint32_t>(r.specularMaxAccumulatedFrameNum, 63);
    p.maxFastASpec = (float)std::min<uint32_t>(r.specularMaxFastAccumulatedFrameNum, 63);
    auto sat01 = [](float v) { return std::min(std::max(v, 0.0f), 1.0f); };
    p.hfNormalPower = r.historyFixEdgeStoppingNormalPower;
    p.alAccel = sat01(r.antilagSettings.accelerationAmount);
    p.alSpatial = r.antilagSettings.spatialSigmaScale;
    p.alTemporal = r.antilagSettings.temporalSigmaScale;
    p.alReset = sat01(r.antilagSettings.resetAmount);
    float n = (float)d.nsig;
    float nr = n * (d.sh ? 2.0f : 1.0f); // radiance texels per pixel (SH mode doubles them)
    float sp = d.hasSpec ? 2.0f : 0.0f;
    const float GB = (float)GUIDE_BYTES;
    {
        Dispatch x{"RELAX::ClassifyTiles", "nrd_reblur_classify_tiles", 0, 4 + 4 + GB + 1.0f / 256.0f, {}, {}, nullptr};
        x.read = {enc_slot(RT::IN_VIEWZ), enc_slot(RT::IN_NORMAL_ROUGHNESS)};
        x.written = {P(rb::GUIDE_A + cur), T(rb::TILES)};
        x.allRows = true;
        { auto q = on_all_rows(p, I.resH); x.launch = [q](hipStream_t st) { launch_reblur_classify_tiles(q, st); }; } // (plain grid, no direction: does not take part in the alternation)
        d.dispatches.push_back(x);
    }
    const PrepareMode pm = prepare_mode(d, s);
    if (pm.any)
        push_prepare_dispatch(d, pm, p, "RELAX::PrepareInputs", P(rb::GUIDE_A + cur), tb);
    {
        Dispatch x{"RELAX::PrePass", "nrd_reblur_prepass", (uint16_t)p.reachPre, GB + 8 * nr + 8 * nr + sp, {}, {}, nullptr};
        x.read = {P(rb::GUIDE_A + cur)};
        push_prepass_inputs(d, pm, tb, x.read);
        x.written = {T(rb::TMP1), T(rb::HITTRACK)};
        x.launch = [p](hipStream_t st) { launch_reblur_spatial(p, 0, st); };
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"RELAX::TemporalAccumulation", "nrd_reblur_temporal_accumulation", 0,
                   GB + 8 + GB + 2 + 8 * nr + 8 * nr + 2 * n + 2 * n + sp + 8 * nr + 2 * n + 2 * n + 2 + 4, {}, {}, nullptr};
        x.read = {P(rb::GUIDE_A + cur), P(rb::GUIDE_A + (cur ^ 1)), enc_slot(RT::IN_MV), T(rb::TMP1), P(rb::HIST), P(rb::FAST_A + (cur ^ 1)),
                  P(rb::DATA1_A + (cur ^ 1)), P(rb::STAB_A + (cur ^ 1)), T(rb::HITTRACK)};
        if (c.mixAvail)
            x.read.push_back(enc_slot(RT::IN_DISOCCLUSION_THRESHOLD_MIX));
        x.written = {T(rb::TMP2), P(rb::FAST_A + cur), P(rb::STAB_A + cur), T(rb::DATA1_TMP), T(rb::DATA2)};
        x.reprojected = {P(rb::GUIDE_A + (cur ^ 1)), P(rb::HIST), P(rb::FAST_A + (cur ^ 1)), P(rb::DATA1_A + (cur ^ 1)), P(rb::STAB_A + (cur ^ 1))};
        x.launch = [p](hipStream_t st) { launch_reblur_temporal_accumulation(p, st); };
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"RELAX::HistoryFix", "nrd_reblur_history_fix", (uint16_t)(2 * s.historyFixBasePixelStride + 2),
                   GB + 2 + 8 * nr + 2 * n + 2 * n + 8 * nr + 2, {}, {}, nullptr};
        x.read = {P(rb::GUIDE_A + cur), T(rb::TMP2), T(rb::DATA1_TMP), P(rb::FAST_A + cur), P(rb::STAB_A + cur)}; // moments: antilag
        x.reach = {{P(rb::FAST_A + cur), (uint16_t)2}};
        x.own = {P(rb::STAB_A + cur)};
        x.written = {P(rb::HIST), P(rb::DATA1_A + cur)};
        { auto q = directed(p, NRD_REVERSE_HISTORY_FIX != 0); x.launch = [q](hipStream_t st) { launch_reblur_history_fix(q, st); }; }
        d.dispatches.push_back(x);
    }
    AtrousParams a;
    std::memset(&a, 0, sizeof(a));
    a.c = p.c; // (with the tile flags of this denoiser's ClassifyTiles)
    a.roughLut = d.roughLut;
    a.depthSens = std::max(r.depthThreshold, 0.001f) * 4.0f;
    a.histThreshold = (float)r.spatialVarianceEstimationHistoryThreshold;
    a.specularVarianceBoost = r.specularVarianceBoost;
    a.phi[0] = r.diffusePhiLuminance;
    a.phi[1] = r.specularPhiLuminance;
    a.minLw[0] = r.diffuseMinLuminanceWeight;
    a.minLw[1] = r.specularMinLuminanceWeight;
    a.lobeAngleFraction = r.lobeAngleFraction;
    a.roughnessFraction = r.roughnessFraction;
    a.minMatDiff = r.minMaterialForDiffuse;
    a.minMatSpec = r.minMaterialForSpecular;
    a.roughnessEdgeStopping = r.enableRoughnessEdgeStopping ? 1 : 0;
    a.lumRelax = sat01(r.luminanceEdgeStoppingRelaxation);
    a.normRelax = sat01(r.normalEdgeStoppingRelaxation);
    a.roughRelax = sat01(r.roughnessEdgeStoppingRelaxation);
    a.data2 = p.data2;
    a.lobeSlack = r.specularLobeAngleSlack * 0.017453292f; // degrees -> radians
    a.confDriven = (r.confidenceDrivenRelaxationMultiplier > 0.0f && c.confAvail) ? 1 : 0;
    a.confMult = r.confidenceDrivenRelaxationMultiplier;
    a.confLumRelax = sat01(r.confidenceDrivenLuminanceEdgeStoppingRelaxation);
    a.confNormRelax = sat01(r.confidenceDrivenNormalEdgeStoppingRelaxation);
    a.confD = p.confD;
    a.confS = p.confS;
    a.hasDiff = d.hasDiff;
    a.hasSpec = d.hasSpec;
    a.sh = d.sh ? 1 : 0;
    a.inDiff1 = p.inDiff1;
    a.inSpec1 = p.inSpec1;
    a.outDiff1 = p.outDiff1;
    a.outSpec1 = p.outSpec1;
    a.guide = p.guide;
    a.tiles = p.tiles;
    a.data1 = p.data1;
    a.hist = p.hist;
    a.mom = p.stab;
    a.inDiff = p.inDiff;
    a.inSpec = p.inSpec;
    a.outDiff = p.outDiff;
    a.outSpec = p.outSpec;
    int iters = (int)std::min<uint32_t>(std::max<uint32_t>(r.atrousIterationNum, 2), 8);
    for (int it = 0; it < iters; it++) {
        bool last = it == iters - 1;
        a.it = it;
        a.last = last ? 1 : 0;
        a.in = it == 0 ? p.hist : I.trans[tb + rb::AT_A + ((it - 1) & 1)].ref();
        a.out = I.trans[tb + rb::AT_A + (it & 1)].ref();
        static const char* atrousNames[8] = {"RELAX::Atrous0", "RELAX::Atrous1", "RELAX::Atrous2", "RELAX::Atrous3", "RELAX::Atrous4", "RELAX::Atrous5", "RELAX::Atrous6", "RELAX::Atrous7"};
        Dispatch x{atrousNames[it], "nrd_relax_atrous", (uint16_t)(1 << it),
                   GB + (it == 0 ? 2 + 8 * nr + 2 * n : 8 * nr) + (last ? 8 * nr : 0.0f) + 8 * nr + ((it <= 2 && d.hasSpec) ? 4.0f : 0.0f), {}, {}, nullptr};
        x.read = {P(rb::GUIDE_A + cur)};
        if (it <= 2 && d.hasSpec) // edge-stopping relaxation reads the reprojection confidence
            x.read.push_back(T(rb::DATA2));
        if (it == 0) {
            x.read.push_back(P(rb::DATA1_A + cur));
            x.read.push_back(P(rb::HIST));
            x.read.push_back(P(rb::STAB_A + cur));
        } else
            x.read.push_back(T(rb::AT_A + ((it - 1) & 1)));
        if (last) {
            x.read.push_back(P(rb::HIST));
            push_signal_slots(d, x.written, true);
            push_signal_slots(d, x.read, false);
        } else
            x.written = {T(rb::AT_A + (it & 1))};
        // An iteration reads the whole plane the previous one wrote. When that plane is larger than what the 256 MiB Infinity Cache keeps of
        // it (SH texels at 4K: 266 MB), odd iterations walk the tiles back to front and start in what is still cached: iterations 1-4
        // -3 ... -6 % (RELAX_DIFFUSE_SPECULAR_SH 4K +1.9 %). A 133 MB plane is still there whichever end the reader starts at: no gain,
        // no reversal (profiles/r03_ab_traversal_direction.txt). Iteration 0 reads what the reversed HistoryFix wrote: forward.
        const bool alternate = (uint64_t)I.resW * (uint64_t)I.resH * (uint64_t)(8u * d.nsig * (d.sh ? 2u : 1u)) > (192ull << 20);
        AtrousParams ap = directed(a, NRD_ATROUS_ALTERNATE != 0 && alternate && (it & 1) != 0);
        x.launch = [ap](hipStream_t st) { launch_relax_atrous(ap, st); };
        d.dispatches.push_back(x);
    }
    push_validation_dispatch(I, d, p, "RELAX::Validation", P(rb::GUIDE_A + cur), P(rb::DATA1_A + cur), T(rb::DATA2));
}

void build_sigma(nrdhip_instance& I, DenoiserState& d, const FrameConsts& c) {
    using RT = nrd::ResourceType;
    int cur = (int)(d.frameCounter & 1);
    uint32_t pb = d.permBase, tb = d.transBase;
    auto P = [&](int i) { return enc_perm(pb + i); };
    auto T = [&](int i) { return enc_trans(tb + i); };
    auto PP = [&](int i) { return I.perm[pb + i].ref(); };
    auto TP = [&](int i) { return I.trans[tb + i].ref(); };
    auto SP = [&](RT t) { return I.slots[(size_t)t].ref(); };
    SigmaParams p;
    std::memset(&p, 0, sizeof(p));
    p.c = c;
    p.planeDistanceSensitivity = d.sigma.planeDistanceSensitivity;
    p.maxStab = (float)std::min<uint32_t>(d.sigma.maxStabilizedFrameNum, nrd::SIGMA_MAX_HISTORY_FRAME_NUM);
    p.translucency = d.translucency ? 1 : 0;
    p.outBpt = (int)I.slots[(size_t)RT::OUT_SHADOW_TRANSLUCENCY].bpt;
    p.inZ = SP(RT::IN_VIEWZ);
    p.inNR = SP(RT::IN_NORMAL_ROUGHNESS);
    p.inMV = SP(RT::IN_MV);
    p.inMix = SP(RT::IN_DISOCCLUSION_THRESHOLD_MIX);
    p.inPen = SP(RT::IN_PENUMBRA);
    p.inTransl = SP(RT::IN_TRANSLUCENCY);
    p.out = SP(RT::OUT_SHADOW_TRANSLUCENCY);
    p.guide = PP(sg::GUIDE_A + cur);
    p.guidePrev = PP(sg::GUIDE_A + (cur ^ 1));
    p.hist = PP(sg::HIST_A + cur);
    p.histPrev = PP(sg::HIST_A + (cur ^ 1));
    p.tiles = TP(sg::TILES);
    p.tilesSmooth = TP(sg::TILES_SMOOTH);
    p.shadow1 = TP(sg::SHADOW1);
    p.pen1 = TP(sg::PEN1);
    p.shadow2 = TP(sg::SHADOW2);
    float tr = d.translucency ? 4.0f : 0.0f;
    const float GB = (float)GUIDE_BYTES;
    {
        Dispatch x{"SIGMA::ClassifyTiles", "nrd_sigma_classify_tiles", 0, 4 + 4 + 2 + GB + 2.0f / 256.0f, {}, {}, nullptr};
        x.read = {enc_slot(RT::IN_VIEWZ), enc_slot(RT::IN_NORMAL_ROUGHNESS), enc_slot(RT::IN_PENUMBRA)};
        x.written = {P(sg::GUIDE_A + cur), T(sg::TILES)};
        x.allRows = true;
        { auto q = on_all_rows(p, I.resH); x.launch = [q](hipStream_t st) { launch_sigma_classify_tiles(q, st); }; }
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"SIGMA::SmoothTiles", "nrd_sigma_smooth_tiles", 16, 4.0f / 256.0f, {}, {}, nullptr};
        x.read = {T(sg::TILES)};
        x.written = {T(sg::TILES_SMOOTH)};
        x.launch = [p](hipStream_t st) { launch_sigma_smooth_tiles(p, st); };
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"SIGMA::Blur", "nrd_sigma_blur", 56, GB + 2 + tr + 8 + 2, {}, {}, nullptr};
        x.read = {P(sg::GUIDE_A + cur), T(sg::TILES_SMOOTH), enc_slot(RT::IN_PENUMBRA)};
        if (d.translucency)
            x.read.push_back(enc_slot(RT::IN_TRANSLUCENCY));
        x.written = {T(sg::SHADOW1), T(sg::PEN1)};
        x.launch = [p](hipStream_t st) { launch_sigma_blur(p, 0, st); };
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"SIGMA::PostBlur", "nrd_sigma_post_blur", 56, GB + 8 + 2 + 8, {}, {}, nullptr};
        x.read = {P(sg::GUIDE_A + cur), T(sg::TILES_SMOOTH), T(sg::SHADOW1), T(sg::PEN1)};
        x.written = {T(sg::SHADOW2)};
        x.launch = [p](hipStream_t st) { launch_sigma_blur(p, 1, st); };
        d.dispatches.push_back(x);
    }
    {
        Dispatch x{"SIGMA::TemporalStabilization", "nrd_sigma_temporal_stabilization", 2, GB + GB + 8 + 8 + 4 + 4 + 4, {}, {}, nullptr};
        x.read = {P(sg::GUIDE_A + cur), P(sg::GUIDE_A + (cur ^ 1)), P(sg::HIST_A + (cur ^ 1)), T(sg::SHADOW2), T(sg::TILES_SMOOTH), enc_slot(RT::IN_MV), enc_slot(RT::IN_PENUMBRA)};
        if (d.translucency)
            x.read.push_back(enc_slot(RT::IN_TRANSLUCENCY));
        if (c.mixAvail)
            x.read.push_back(enc_slot(RT::IN_DISOCCLUSION_THRESHOLD_MIX));
        x.written = {P(sg::HIST_A + cur), enc_slot(RT::OUT_SHADOW_TRANSLUCENCY)};
        x.reprojected = {P(sg::GUIDE_A + (cur ^ 1)), P(sg::HIST_A + (cur ^ 1))};
        x.launch = [p](hipStream_t st) { launch_sigma_temporal_stabilization(p, st); };
        d.dispatches.push_back(x);
    }
}

struct Flat {
    DenoiserState* d;
    uint32_t index;
};

// The tile order of a launch over tilesX x tilesY tiles (nrd_device.h xcd_tile_kj, forward direction, tile rows relative to tileY0) as a
// device table: entry j * 8 + k = tx | ty << 16 of the j-th tile of XCD k, 0xffffffff for the spare workgroups of the rounded-up grid.
// nullptr when the shape has no table and none may be made now (inside a stream capture) or the allocation fails: the kernels then
// compute the tile themselves - same order, same result.
// The upload is stream-ordered on the stream the kernels are launched on (no device-wide synchronisation in the middle of a row tiler's
// overlapped exchange schedule).
const uint32_t* tile_table(nrdhip_instance& I, int tilesX, int tilesY, hipStream_t st) {
    if (tilesX <= 0 || tilesY <= 0 || tilesX > 0xffff || tilesY > 0xfffe)
        return nullptr;
    auto it = I.tileTables.find({tilesX, tilesY});
    if (it != I.tileTables.end())
        return it->second.dev;
    if (I.capturing || stream_is_capturing(st))
        return nullptr; // (inside the library's OR THE CALLER'S stream capture nothing may be allocated or copied; not remembered: the next plain launch builds it)
    const int blocks = xcd_grid_blocks(tilesX, tilesY);
    if (I.tileTables.size() >= 64)
        return nullptr; // (a caller resizing the rect every frame: the shapes beyond the first 64 compute their tiles)
    nrdhip_instance::TileTable& T = I.tileTables[{tilesX, tilesY}];
    std::vector<uint32_t>& host = T.host;
    host.assign((size_t)blocks + (size_t)tilesX * tilesY, 0xffffffffu);
    T.blocks = (uint32_t)blocks;
    FrameConsts c;
    std::memset(&c, 0, sizeof(c));
    c.tilesX = tilesX;
    c.tilesY = tilesY;
    for (int b = 0; b < blocks; b++) {
        int tx, ty;
        if (xcd_tile_kj(c, b & 7, b >> 3, tx, ty)) {
            host[(size_t)b] = (uint32_t)tx | ((uint32_t)ty << 16);
            host[(size_t)blocks + (size_t)ty * tilesX + tx] = (uint32_t)b;
        }
    }
    // a BLOCKING copy, once per shape for the life of the instance: the table is then complete for every stream that will ever read it (ADVICE
    // r5: an asynchronous upload is ordered on the creating stream only - a strip launched on another stream could read it half-written)
    uint32_t* dev = nullptr;
    if (hipMalloc((void**)&dev, host.size() * sizeof(uint32_t)) != hipSuccess ||
        hipMemcpy(dev, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        if (dev)
            (void)hipFree(dev);
        dev = nullptr; // (remembered: the shape is not tried again every frame)
    }
    T.dev = dev;
    return dev;
}

// forLaunch: the lists are about to be launched (the HIP device of the instance is current): their grids get their tile tables
int flatten(nrdhip_instance& I, const uint32_t* ids, uint32_t n, std::vector<Flat>& out, bool forLaunch = false, hipStream_t st = nullptr) {
    out.clear();
    if (!I.commonSet) {
        I.error = "SetCommonSettings has not been called";
        return (int)nrd::Result::INVALID_ARGUMENT;
    }
    FrameConsts c;
    if (!derive_consts(I, c, I.error))
        return (int)nrd::Result::INVALID_ARGUMENT;
    if (forLaunch) {
        I.tableStream = st;
        c.tileTable = tile_table(I, c.tilesX, c.tilesY, st);
        c.tilesPerXcd = xcd_grid_blocks(c.tilesX, c.tilesY) / 8;
    }
    bool reset = I.common.accumulationMode != nrd::AccumulationMode::CONTINUE;
    for (uint32_t i = 0; i < n; i++) {
        DenoiserState* d = find(I, ids[i]);
        if (!d) {
            I.error = "unknown identifier";
            return (int)nrd::Result::INVALID_ARGUMENT;
        }
        d->dispatches.clear();
        c.historyOk = (d->historyValid && !reset) ? 1 : 0;
        switch (d->kind) {
            case Kind::REFERENCE: build_reference(I, *d, c); break;
            case Kind::REBLUR: build_reblur(I, *d, c); break;
            case Kind::SIGMA: build_sigma(I, *d, c); break;
            case Kind::RELAX: build_relax(I, *d, c); break;
        }
        for (uint32_t k = 0; k < d->dispatches.size(); k++)
            out.push_back({d, k});
    }
    return 0;
}

} // namespace

// Stage annotations, like the reference's nri Annotation scopes around every stage (Source/NRDSample.cpp:4070, :4087, :4214): one roctx
// range per recorded dispatch, named after the pass ("REBLUR::Blur" ...), visible in `rocprofv3 --marker-trace`. libroctx64 is
// resolved on first use (no link dependency); NRDHIP_MARKERS=0 turns the ranges off.
struct Markers {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Markers() {
        const char* env = std::getenv("NRDHIP_MARKERS");
        if (env && env[0] == '0')
            return;
        void* lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!lib)
            lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib)
            return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!push || !pop)
            push = nullptr, pop = nullptr;
    }
};
struct MarkerScope {
    const Markers& m;
    MarkerScope(const Markers& markers, const char* name) : m(markers) {
        if (m.push)
            m.push(name);
    }
    ~MarkerScope() {
        if (m.pop)
            m.pop();
    }
};
const Markers& markers() {
    static Markers m;
    return m;
}

// makes the instance's device current for the duration of a call (Recreate(..., device) of the C++ veneer), restores the caller's
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int device) {
        if (device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != device)
            switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope() {
        if (switched)
            (void)hipSetDevice(prev);
    }
};

// the instances alive in this process: a row tiler that is destroyed AFTER its instance (destructor order of a host language) must
// not touch it (nrdhip_tiler.cpp nrdhip_tiler_destroy)
static std::mutex g_liveMutex;
static std::vector<const nrdhip_instance*> g_live;
namespace nrdhip {
bool instance_is_live(const nrdhip_instance* inst) {
    std::lock_guard<std::mutex> lock(g_liveMutex);
    return std::find(g_live.begin(), g_live.end(), inst) != g_live.end();
}
} // namespace nrdhip

extern "C" {

NRDHIP_API int nrdhip_create(const nrdhip_create_desc* desc, nrdhip_instance** out) {
    if (!desc || !out || !desc->denoisers || !desc->denoisers_num || !desc->resource_width || !desc->resource_height) {
        g_createError = "invalid creation desc";
        return (int)nrd::Result::INVALID_ARGUMENT;
    }
    auto* I = new nrdhip_instance();
    I->resW = desc->resource_width;
    I->resH = desc->resource_height;
    I->frameH = desc->frame_height ? desc->frame_height : desc->resource_height;
    I->yOff = desc->band_row0;
    I->ownY0 = desc->band_own_first;
    I->ownRows = desc->band_own_rows;
    I->flags = desc->flags;
    I->device = desc->device_plus1 ? (int)desc->device_plus1 - 1 : -1;
    if (I->device >= 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || I->device >= count) {
            g_createError = "no such HIP device";
            delete I;
            return (int)nrd::Result::INVALID_ARGUMENT;
        }
    }
    DeviceScope scope(I->device);
    std::vector<PoolPlane> permDesc, transDesc;
    for (uint32_t i = 0; i < desc->denoisers_num; i++) {
        DenoiserState d;
        d.identifier = desc->denoisers[i].identifier;
        if (find(*I, d.identifier)) {
            g_createError = "non unique identifier";
            nrdhip_destroy(I); // (frees what the denoisers in front of this one hold)
            return (int)nrd::Result::NON_UNIQUE_IDENTIFIER;
        }
        if (desc->denoisers[i].denoiser >= (uint32_t)nrd::Denoiser::MAX_NUM || !classify((nrd::Denoiser)desc->denoisers[i].denoiser, d)) {
            g_createError = "unsupported denoiser";
            nrdhip_destroy(I);
            return (int)nrd::Result::UNSUPPORTED;
        }
        d.permBase = (uint32_t)permDesc.size();
        d.transBase = (uint32_t)transDesc.size();
        describe(d, permDesc, transDesc);
        d.permEnd = (uint32_t)permDesc.size();
        if ((d.kind == Kind::REBLUR || d.kind == Kind::RELAX) && d.hasSpec) {
            if (hipMalloc((void**)&d.roughLut, NRDHIP_ROUGH_LUT_FLOATS * sizeof(float)) != hipSuccess || hipMemset(d.roughLut, 0, NRDHIP_ROUGH_LUT_FLOATS * sizeof(float)) != hipSuccess) {
                (void)hipGetLastError();
                g_createError = "hipMalloc failed (is a HIP device visible?)";
                I->denoisers.push_back(d);
                nrdhip_destroy(I);
                return (int)nrd::Result::FAILURE;
            }
        }
        I->denoisers.push_back(d);
    }
    auto make = [&](std::vector<PoolPlane>& descs, std::vector<Plane>& planes) -> bool {
        for (auto& pd : descs) {
            Plane P;
            P.fmt = (uint32_t)pd.fmt;
            P.bpt = pd.bpt;
            P.name = pd.name;
            P.w = (uint16_t)((I->resW + pd.downsample - 1) / pd.downsample);
            P.h = (uint16_t)((I->resH + pd.downsample - 1) / pd.downsample);
            P.pitch = P.w * P.bpt;
            if (!(I->flags & NRDHIP_FLAG_EXTERNAL_POOLS)) {
                size_t bytes = (size_t)P.pitch * P.h;
                if (hipMalloc((void**)&P.p, bytes) != hipSuccess)
                    return false;
                (void)hipMemset(P.p, 0, bytes);
                P.owned = true;
            }
            planes.push_back(P);
        }
        return true;
    };
    // Transient planes do not survive a frame and the denoisers of an instance run one after the other on one stream, so the
    // library-owned transient pool is ONE arena sized for the hungriest denoiser and every denoiser's transient planes alias into it
    // (what Get AliasableMemoryUsageInMb of the reference reports as aliasable IS aliased here). Callers that own the pools
    // (NRDHIP_FLAG_EXTERNAL_POOLS) get one description per plane and may alias them the same way.
    auto make_transient_arena = [&]() -> bool {
        const bool owned = !(I->flags & NRDHIP_FLAG_EXTERNAL_POOLS);
        size_t arena = 0;
        std::vector<size_t> offsets(transDesc.size(), 0);
        for (size_t di = 0; di < I->denoisers.size(); di++) {
            const uint32_t b = I->denoisers[di].transBase, e = di + 1 < I->denoisers.size() ? I->denoisers[di + 1].transBase : (uint32_t)transDesc.size();
            size_t off = 0;
            for (uint32_t k = b; k < e; k++) {
                const PoolPlane& pd = transDesc[k];
                size_t w = (size_t)(I->resW + pd.downsample - 1) / pd.downsample, h = (size_t)(I->resH + pd.downsample - 1) / pd.downsample;
                offsets[k] = off;
                off += (w * pd.bpt * h + 255) / 256 * 256; // 256-byte aligned planes
            }
            arena = std::max(arena, off);
        }
        I->transArenaBytes = arena;
        if (owned && arena) {
            if (hipMalloc((void**)&I->transArena, arena) != hipSuccess)
                return false;
            (void)hipMemset(I->transArena, 0, arena);
        }
        for (size_t k = 0; k < transDesc.size(); k++) {
            const PoolPlane& pd = transDesc[k];
            Plane P;
            P.fmt = (uint32_t)pd.fmt;
            P.bpt = pd.bpt;
            P.name = pd.name;
            P.w = (uint16_t)((I->resW + pd.downsample - 1) / pd.downsample);
            P.h = (uint16_t)((I->resH + pd.downsample - 1) / pd.downsample);
            P.pitch = P.w * P.bpt;
            if (owned)
                P.p = I->transArena + offsets[k]; // not `owned`: the arena is freed as a whole
            I->trans.push_back(P);
        }
        return true;
    };
    if (!make(permDesc, I->perm) || !make_transient_arena()) {
        g_createError = "hipMalloc failed (is a HIP device visible?)";
        nrdhip_destroy(I);
        return (int)nrd::Result::FAILURE;
    }
    {
        std::lock_guard<std::mutex> lock(g_liveMutex);
        g_live.push_back(I);
    }
    *out = I;
    return 0;
}

// every executable graph of the instance, each after the launch that used it last has finished
static void release_graphs(nrdhip_instance& I) {
    for (auto& g : I.graphExecs)
        for (int i = 0; i < nrdhip_instance::GraphSlot::RING; i++) {
            if (g.done[i]) {
                (void)hipEventSynchronize(g.done[i]);
                (void)hipEventDestroy(g.done[i]);
            }
            if (g.exec[i])
                (void)hipGraphExecDestroy(g.exec[i]);
        }
    I.graphExecs.clear();
}

NRDHIP_API void nrdhip_destroy(nrdhip_instance* inst) {
    if (!inst)
        return;
    DeviceScope scope(inst->device);
    for (auto* v : {&inst->perm, &inst->trans})
        for (auto& P : *v)
            if (P.owned && P.p)
                (void)hipFree(P.p);
    if (inst->transArena)
        (void)hipFree(inst->transArena);
    for (auto& t : inst->tileTables)
        if (t.second.dev)
            (void)hipFree(t.second.dev);
    for (auto& d : inst->denoisers) {
        for (auto& f : d.tileFlags)
            if (f.second)
                (void)hipFree(f.second);
        if (d.roughLut)
            (void)hipFree(d.roughLut);
    }
    release_graphs(*inst);
    {
        std::lock_guard<std::mutex> lock(g_liveMutex);
        g_live.erase(std::remove(g_live.begin(), g_live.end(), (const nrdhip_instance*)inst), g_live.end());
    }
    delete inst;
}

NRDHIP_API int nrdhip_new_frame(nrdhip_instance* inst) { return inst ? 0 : (int)nrd::Result::INVALID_ARGUMENT; }

NRDHIP_API int nrdhip_set_common(nrdhip_instance* inst, const void* settings, size_t size) {
    if (!inst || !settings || size != sizeof(nrd::CommonSettings))
        return (int)nrd::Result::INVALID_ARGUMENT;
    std::memcpy(&inst->common, settings, size);
    inst->commonSet = true;
    return 0;
}

NRDHIP_API int nrdhip_set_denoiser(nrdhip_instance* inst, uint32_t identifier, const void* settings, size_t size) {
    if (!inst || !settings)
        return (int)nrd::Result::INVALID_ARGUMENT;
    DenoiserState* d = find(*inst, identifier);
    if (!d)
        return (int)nrd::Result::INVALID_ARGUMENT;
    void* dst = nullptr;
    size_t want = 0;
    switch (d->kind) {
        case Kind::REBLUR: dst = &d->reblur; want = sizeof(d->reblur); break;
        case Kind::RELAX: dst = &d->relax; want = sizeof(d->relax); break;
        case Kind::SIGMA: dst = &d->sigma; want = sizeof(d->sigma); break;
        case Kind::REFERENCE: dst = &d->reference; want = sizeof(d->reference); break;
    }
    if (size != want)
        return (int)nrd::Result::INVALID_ARGUMENT;
    std::memcpy(dst, settings, size);
    return 0;
}

NRDHIP_API int nrdhip_bind(nrdhip_instance* inst, uint32_t slot, void* ptr, uint32_t pitch, uint32_t format, uint16_t w, uint16_t h) {
    if (!inst || slot >= (uint32_t)nrd::ResourceType::TRANSIENT_POOL)
        return (int)nrd::Result::INVALID_ARGUMENT;
    Plane& P = inst->slots[slot];
    P.p = (uint8_t*)ptr;
    P.pitch = pitch;
    P.fmt = format;
    P.w = w;
    P.h = h;
    P.bpt = format_bytes(format);
    return 0;
}

NRDHIP_API int nrdhip_unbind_all(nrdhip_instance* inst) {
    if (!inst)
        return (int)nrd::Result::INVALID_ARGUMENT;
    for (auto& P : inst->slots)
        P = Plane();
    return 0;
}

NRDHIP_API int nrdhip_denoiser_kind(nrdhip_instance* inst, uint32_t identifier, uint32_t* kind) {
    DenoiserState* d = inst ? find(*inst, identifier) : nullptr;
    if (!d || !kind)
        return (int)nrd::Result::INVALID_ARGUMENT;
    *kind = (uint32_t)d->kind;
    return 0;
}

NRDHIP_API int nrdhip_get_device(nrdhip_instance* inst) { return inst ? inst->device : -1; }

NRDHIP_API int nrdhip_set_history_rows(nrdhip_instance* inst, int32_t first_local_row, uint32_t rows) {
    if (!inst)
        return (int)nrd::Result::INVALID_ARGUMENT;
    inst->histY0 = first_local_row;
    inst->histRows = (int)rows;
    return 0;
}

NRDHIP_API int nrdhip_get_band(nrdhip_instance* inst, int32_t out[5]) {
    if (!inst || !out)
        return (int)nrd::Result::INVALID_ARGUMENT;
    out[0] = inst->frameH;
    out[1] = inst->yOff;
    out[2] = inst->ownY0;
    out[3] = inst->ownRows ? inst->ownRows : inst->resH - inst->ownY0;
    out[4] = inst->resH;
    return 0;
}

NRDHIP_API int nrdhip_slot_info(nrdhip_instance* inst, uint32_t slot, nrdhip_plane_info* out) {
    if (!inst || !out || slot >= (uint32_t)nrd::ResourceType::TRANSIENT_POOL)
        return (int)nrd::Result::INVALID_ARGUMENT;
    const Plane& P = inst->slots[slot];
    out->ptr = P.p;
    out->pitch_bytes = P.pitch;
    out->format = P.fmt;
    out->width = P.w;
    out->height = P.h;
    out->bytes_per_texel = P.bpt;
    out->name = "slot";
    return 0;
}

NRDHIP_API int nrdhip_pool_size(nrdhip_instance* inst, uint32_t pool, uint32_t* count) {
    if (!inst || !count || pool > 1)
        return (int)nrd::Result::INVALID_ARGUMENT;
    *count = (uint32_t)(pool == 0 ? inst->perm.size() : inst->trans.size());
    return 0;
}

NRDHIP_API int nrdhip_pool_info(nrdhip_instance* inst, uint32_t pool, uint32_t index, nrdhip_plane_info* out) {
    if (!inst || !out || pool > 1)
        return (int)nrd::Result::INVALID_ARGUMENT;
    auto& v = pool == 0 ? inst->perm : inst->trans;
    if (index >= v.size())
        return (int)nrd::Result::INVALID_ARGUMENT;
    const Plane& P = v[index];
    out->ptr = P.p;
    out->pitch_bytes = P.pitch;
    out->format = P.fmt;
    out->width = P.w;
    out->height = P.h;
    out->bytes_per_texel = P.bpt;
    out->name = P.name;
    return 0;
}

NRDHIP_API int nrdhip_bind_pool(nrdhip_instance* inst, uint32_t pool, uint32_t index, void* ptr, uint32_t pitch) {
    if (!inst || pool > 1)
        return (int)nrd::Result::INVALID_ARGUMENT;
    auto& v = pool == 0 ? inst->perm : inst->trans;
    if (index >= v.size() || pitch < v[index].w * v[index].bpt || v[index].owned || !(inst->flags & NRDHIP_FLAG_EXTERNAL_POOLS))
        return (int)nrd::Result::INVALID_ARGUMENT;
    v[index].p = (uint8_t*)ptr;
    v[index].pitch = pitch;
    return 0;
}

NRDHIP_API int nrdhip_dispatch_count(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t* count) {
    if (!inst || !count)
        return (int)nrd::Result::INVALID_ARGUMENT;
    std::vector<Flat> fl;
    int r = flatten(*inst, ids, n, fl);
    *count = (uint32_t)fl.size();
    return r;
}

NRDHIP_API int nrdhip_dispatch_info_get(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t index, nrdhip_dispatch_info* out) {
    if (!inst || !out)
        return (int)nrd::Result::INVALID_ARGUMENT;
    std::vector<Flat> fl;
    int r = flatten(*inst, ids, n, fl);
    if (r)
        return r;
    if (index >= fl.size())
        return (int)nrd::Result::INVALID_ARGUMENT;
    const Dispatch& x = fl[index].d->dispatches[fl[index].index];
    std::memset(out, 0, sizeof(*out));
    out->name = x.name;
    out->kernel = x.kernel;
    out->identifier = fl[index].d->identifier;
    out->grid_width = (uint16_t)((inst->common.rectSize[0] + 15) / 16);
    out->grid_height = (uint16_t)((inst->common.rectSize[1] + 15) / 16);
    out->halo_rows = x.halo;
    out->written_num = (uint16_t)std::min<size_t>(x.written.size(), 12);
    for (uint32_t i = 0; i < out->written_num; i++)
        out->written[i] = x.written[i];
    out->read_num = (uint32_t)std::min<size_t>(x.read.size(), 24);
    for (uint32_t i = 0; i < out->read_num; i++) {
        out->read[i] = x.read[i];
        uint16_t rows = x.halo;
        if (std::find(x.own.begin(), x.own.end(), x.read[i]) != x.own.end())
            rows = 0;
        if (std::find(x.reprojected.begin(), x.reprojected.end(), x.read[i]) != x.reprojected.end())
            rows = (uint16_t)NRDHIP_READ_REPROJECTED;
        for (auto& rc : x.reach)
            if (rc.first == x.read[i])
                rows = std::min(rc.second, x.halo);
        out->read_rows[i] = rows;
    }
    out->flags = x.allRows ? (uint32_t)NRDHIP_DISPATCH_ALL_ROWS : 0u;
    for (uint32_t i = 0; i < 12; i++) {
        out->written_prefix[i] = (uint32_t)NRDHIP_NO_PLANE;
        for (auto& pf : x.prefix)
            if (i < out->written_num && pf.first == out->written[i])
                out->written_prefix[i] = pf.second;
    }
    out->algorithmic_bytes_per_pixel = x.bpp;
    return 0;
}

static int denoise_parts(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t first, uint32_t count, uint32_t part, void* stream);

NRDHIP_API int nrdhip_denoise_range(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t first, uint32_t count, void* stream) {
    return denoise_parts(inst, ids, n, first, count, NRDHIP_PART_FIRST | NRDHIP_PART_LAST, stream);
}

NRDHIP_API int nrdhip_denoise_rows(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t index, uint32_t row_first, uint32_t row_count,
                                   uint32_t part, void* stream) {
    if (!inst)
        return (int)nrd::Result::INVALID_ARGUMENT;
    nrdhip_instance& I = *inst;
    // clip the requested window to the rows this instance owns, run the dispatch on it, restore
    const int own0 = I.ownY0, ownN = I.ownRows;
    int lo = std::max((int)row_first, own0), hi = (int)(row_first + row_count);
    if (ownN)
        hi = std::min(hi, own0 + ownN);
    hi = std::min(hi, (int)I.resH);
    int r = 0;
    if (hi > lo) {
        I.ownY0 = lo;
        I.ownRows = hi - lo;
        r = denoise_parts(inst, ids, n, index, 1, part, stream);
        I.ownY0 = own0;
        I.ownRows = ownN;
    } else if (part & (NRDHIP_PART_FIRST | NRDHIP_PART_LAST)) {
        // nothing to compute (a strip outside the rect under DRS), but the bookkeeping of the part must still happen: FIRST issues
        // the CLEAR_AND_RESTART clear of the permanent pool, LAST advances the frame state
        r = denoise_parts(inst, ids, n, index, 1, part | 4u, stream);
    }
    return r;
}

// part: NRDHIP_PART_FIRST / NRDHIP_PART_LAST, bit 2 = bookkeeping only (no launch)
static int denoise_parts(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t first, uint32_t count, uint32_t part, void* stream) {
    if (!inst)
        return (int)nrd::Result::INVALID_ARGUMENT;
    nrdhip_instance& I = *inst;
    hipStream_t st = (hipStream_t)stream;
    DeviceScope scope(I.device);
    for (auto* v : {&I.perm, &I.trans})
        for (auto& P : *v)
            if (!P.p) {
                I.error = "pool plane not bound";
                return (int)nrd::Result::INVALID_ARGUMENT;
            }
    std::vector<Flat> fl;
    int r = flatten(I, ids, n, fl, true, st);
    if (r)
        return r;
    if (first > fl.size() || first + count > fl.size())
        return (int)nrd::Result::INVALID_ARGUMENT;
    bool reset = I.common.accumulationMode != nrd::AccumulationMode::CONTINUE;
    for (uint32_t i = first; i < first + count; i++) {
        DenoiserState& d = *fl[i].d;
        Dispatch& x = d.dispatches[fl[i].index];
        for (auto* lst : {&x.read, &x.written})
            for (uint32_t s : *lst)
                if ((s >> 16) == 2 && !I.slots[s & 0xffff].p) {
                    // confidence inputs are optional (sampled as 1 when absent)
                    if ((s & 0xffff) == (uint32_t)nrd::ResourceType::IN_DIFF_CONFIDENCE || (s & 0xffff) == (uint32_t)nrd::ResourceType::IN_SPEC_CONFIDENCE)
                        continue;
                    I.error = std::string("resource slot not bound for pass ") + x.name;
                    return (int)nrd::Result::INVALID_ARGUMENT;
                } else if ((s >> 16) == 2 && I.slots[s & 0xffff].p && !nrd::IsFormatAllowed((nrd::ResourceType)(s & 0xffff), (nrd::Format)I.slots[s & 0xffff].fmt)) {
                    I.error = std::string("resource slot bound with an unsupported format for pass ") + x.name;
                    return (int)nrd::Result::INVALID_ARGUMENT;
                } else if ((s >> 16) == 2 && I.slots[s & 0xffff].p) {
                    // the kernels index a bound plane over the rect (rows: the rows this instance holds): a smaller plane or a pitch
                    // shorter than its texels would be read / written out of bounds. Confidence planes are sampled by uv at their
                    // own resolution; checkerboarded inputs are half width.
                    const uint32_t slot = s & 0xffff;
                    const Plane& P = I.slots[slot];
                    const bool conf = slot == (uint32_t)nrd::ResourceType::IN_DIFF_CONFIDENCE || slot == (uint32_t)nrd::ResourceType::IN_SPEC_CONFIDENCE;
                    // only the noisy signal inputs of a checkerboarded denoiser are half width; every other slot spans the rect
                    bool halfWidth = false;
                    if (d.kind == Kind::REBLUR || d.kind == Kind::RELAX) {
                        const bool checker = (d.kind == Kind::REBLUR ? d.reblur.checkerboardMode : d.relax.checkerboardMode) != nrd::CheckerboardMode::OFF;
                        using RT = nrd::ResourceType;
                        const RT t = (RT)slot;
                        halfWidth = checker && (t == in_slot(d, false) || t == in_slot(d, true) || t == RT::IN_DIFF_SH1 || t == RT::IN_SPEC_SH1);
                    }
                    const uint32_t needW = conf ? 1u : (halfWidth ? (uint32_t)(I.common.rectSize[0] + 1) / 2u : (uint32_t)I.common.rectSize[0]);
                    const uint32_t needH = conf ? 1u : (uint32_t)std::min<int>(I.resH, std::max<int>((int)I.common.rectSize[1] - I.yOff, 1));
                    if (P.pitch < (uint32_t)P.w * P.bpt || P.w < needW || P.h < needH) {
                        I.error = std::string("resource slot plane smaller than the rect (or pitch < width x texel size) for pass ") + x.name;
                        return (int)nrd::Result::INVALID_ARGUMENT;
                    }
                }
        // texel offsets are 32-bit, and the tap loops address band planes by GLOBAL row (the band's first row is folded into
        // the base pointer): (first row + rows held) x pitch must stay below 4 GiB for every plane the pass touches
        for (auto* lst : {&x.read, &x.written})
            for (uint32_t s : *lst) {
                const Plane* P = (s >> 16) == 0 ? &I.perm[s & 0xffff] : ((s >> 16) == 1 ? &I.trans[s & 0xffff] : ((s >> 16) == 2 ? &I.slots[s & 0xffff] : nullptr));
                if (P && P->p && ((uint64_t)I.yOff + (uint64_t)I.resH) * (uint64_t)P->pitch > 0xffffffffull) {
                    I.error = std::string("plane too large for 32-bit texel offsets in pass ") + x.name;
                    return (int)nrd::Result::UNSUPPORTED;
                }
            }
        // (also before the very first frame of a denoiser, whatever the mode: passes leave the tiles without geometry unwritten, and a history
        // texel nobody has ever written must still hold a finite value - caller-allocated pools arrive with whatever the allocator left)
        if ((part & NRDHIP_PART_FIRST) && fl[i].index == 0 && (I.common.accumulationMode == nrd::AccumulationMode::CLEAR_AND_RESTART || !d.historyValid))
            for (uint32_t k = d.permBase; k < d.permEnd; k++)
                (void)hipMemset2DAsync(I.perm[k].p, I.perm[k].pitch, 0, (size_t)I.perm[k].w * I.perm[k].bpt, I.perm[k].h, st);
        if (!(part & 4u)) {
            MarkerScope scope(markers(), x.name);
            x.launch(st);
        }
        if ((part & NRDHIP_PART_LAST) && fl[i].index + 1 == d.dispatches.size()) {
            d.framesSinceReset = (reset || !d.historyValid) ? 1 : d.framesSinceReset + 1;
            d.frameCounter++;
            d.historyValid = true;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        I.error = std::string("HIP launch failed: ") + hipGetErrorString(e);
        return (int)nrd::Result::FAILURE;
    }
    return 0;
}

NRDHIP_API int nrdhip_denoise(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, void* stream) {
    uint32_t count = 0;
    int r = nrdhip_dispatch_count(inst, ids, n, &count);
    if (r)
        return r;
    if (!(inst->flags & NRDHIP_FLAG_GRAPH))
        return nrdhip_denoise_range(inst, ids, n, 0, count, stream);
    // One graph launch per frame. The kernels take their per-frame constants by value, so the list is captured again every frame -
    // capture records nodes, it launches nothing - and the executable graph of the previous frame is patched with the new arguments
    nrdhip_instance& I = *inst;
    hipStream_t st = (hipStream_t)stream;
    DeviceScope scope(I.device);
    { // the tile table of the frame's grid is made BEFORE the capture begins (an allocation + copy; nothing of the kind may run inside it)
        FrameConsts c;
        std::string err;
        if (I.commonSet && derive_consts(I, c, err)) {
            I.tableStream = st;
            c.tileTable = tile_table(I, c.tilesX, c.tilesY, st);
            c.tilesPerXcd = xcd_grid_blocks(c.tilesX, c.tilesY) / 8;
            // ... and the launch-order flag buffers of the REBLUR / RELAX denoisers of this list (attach_tile_flags allocates on first use)
            for (uint32_t i = 0; i < n; i++) {
                DenoiserState* d = find(I, ids[i]);
                if (d && (d->kind == Kind::REBLUR || d->kind == Kind::RELAX)) {
                    ReblurParams q;
                    std::memset(&q, 0, sizeof(q));
                    q.c = c;
                    attach_tile_flags(I, *d, q);
                }
            }
        }
    }
    struct CaptureFlag {
        bool& f;
        explicit CaptureFlag(bool& b) : f(b) { f = true; }
        ~CaptureFlag() { f = false; }
    };
    if (!st || hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        I.graphStats[2]++;
        return nrdhip_denoise_range(inst, ids, n, 0, count, stream);
    }
    // the capture pass below advances the denoisers' frame counters / ping-pong parity / first-frame flags although no GPU work runs yet:
    // snapshot them, so that a failure anywhere between capture and launch can hand the SAME frame to the direct path
    struct Saved {
        DenoiserState* d;
        uint32_t frameCounter, framesSinceReset;
        bool historyValid;
    };
    std::vector<Saved> saved;
    for (auto& d : I.denoisers)
        saved.push_back({&d, d.frameCounter, d.framesSinceReset, d.historyValid});
    auto direct_fallback = [&](const char* what, hipError_t err) {
        (void)hipGetLastError();
        for (auto& sv : saved) {
            sv.d->frameCounter = sv.frameCounter;
            sv.d->framesSinceReset = sv.framesSinceReset;
            sv.d->historyValid = sv.historyValid;
        }
        I.error = std::string(what) + " failed (" + hipGetErrorString(err) + "): frame submitted pass by pass";
        I.graphStats[2]++;
        return nrdhip_denoise_range(inst, ids, n, 0, count, stream);
    };
    {
        CaptureFlag capturing(I.capturing);
        r = nrdhip_denoise_range(inst, ids, n, 0, count, stream);
    }
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (r) { // the dispatch list itself is at fault (unbound slot, ...): the direct path would fail the same way
        if (graph)
            (void)hipGraphDestroy(graph);
        for (auto& sv : saved) {
            sv.d->frameCounter = sv.frameCounter;
            sv.d->framesSinceReset = sv.framesSinceReset;
            sv.d->historyValid = sv.historyValid;
        }
        return r;
    }
    if (e != hipSuccess || !graph) {
        if (graph)
            (void)hipGraphDestroy(graph);
        return direct_fallback("HIP stream capture", e);
    }
    const std::vector<uint32_t> key(ids, ids + n);
    size_t slot = 0;
    while (slot < I.graphExecs.size() && I.graphExecs[slot].key != key)
        slot++;
    if (slot == I.graphExecs.size()) {
        if (I.graphExecs.size() >= 16) { // (a caller cycling through many identifier lists: start over)
            release_graphs(I);
            slot = 0;
        }
        I.graphExecs.emplace_back();
        I.graphExecs.back().key = key;
    }
    nrdhip_instance::GraphSlot& G = I.graphExecs[slot];
    const int ring = (int)(G.next++ % nrdhip_instance::GraphSlot::RING);
    hipGraphExec_t& exec = G.exec[ring];
    if (G.done[ring])
        (void)hipEventSynchronize(G.done[ring]); // the launch that used this executable graph last (RING frames ago) has left the device
    if (exec) {
        hipGraphNode_t bad = nullptr;
        hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
        if (hipGraphExecUpdate(exec, graph, &bad, &res) != hipSuccess || res != hipGraphExecUpdateSuccess) {
            (void)hipGetLastError();
            (void)hipGraphExecDestroy(exec);
            exec = nullptr;
        }
    }
    if (!exec) {
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            exec = nullptr;
            (void)hipGraphDestroy(graph);
            return direct_fallback("hipGraphInstantiate", e);
        }
        I.graphStats[1]++;
    }
    e = hipGraphLaunch(exec, st);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess)
        return direct_fallback("hipGraphLaunch", e);
    if (!G.done[ring] && hipEventCreateWithFlags(&G.done[ring], hipEventDisableTiming) != hipSuccess)
        G.done[ring] = nullptr;
    if (G.done[ring])
        (void)hipEventRecord(G.done[ring], st);
    else
        (void)hipStreamSynchronize(st); // no event to order the next update behind: wait here
    I.graphStats[0]++;
    return 0;
}

NRDHIP_API int nrdhip_get_history_state(nrdhip_instance* inst, uint32_t identifier, nrdhip_history_state* out) {
    DenoiserState* d = inst ? find(*inst, identifier) : nullptr;
    if (!d || !out)
        return (int)nrd::Result::INVALID_ARGUMENT;
    *out = {d->frameCounter, d->framesSinceReset, d->historyValid ? 1u : 0u, 0u};
    return 0;
}

NRDHIP_API int nrdhip_set_history_state(nrdhip_instance* inst, uint32_t identifier, const nrdhip_history_state* in) {
    DenoiserState* d = inst ? find(*inst, identifier) : nullptr;
    if (!d || !in)
        return (int)nrd::Result::INVALID_ARGUMENT;
    d->frameCounter = in->frame_counter;
    d->framesSinceReset = in->frames_since_reset;
    d->historyValid = in->history_valid != 0;
    return 0;
}

NRDHIP_API int nrdhip_graph_stats(nrdhip_instance* inst, uint32_t out[3]) {
    if (!inst || !out)
        return (int)nrd::Result::INVALID_ARGUMENT;
    for (int i = 0; i < 3; i++)
        out[i] = inst->graphStats[i];
    return 0;
}

NRDHIP_API int nrdhip_get_memory_mb(nrdhip_instance* inst, float out[3]) {
    if (!inst || !out)
        return (int)nrd::Result::INVALID_ARGUMENT;
    double p = 0, t = 0;
    for (auto& P : inst->perm)
        p += (double)P.pitch * P.h;
    if (!(inst->flags & NRDHIP_FLAG_EXTERNAL_POOLS)) {
        t = (double)inst->transArenaBytes; // aliased: one arena for the hungriest denoiser
    } else {
        for (auto& P : inst->trans)
            t += (double)P.pitch * P.h;
    }
    out[0] = (float)((p + t) / 1048576.0);
    out[1] = (float)(p / 1048576.0);
    out[2] = (float)(t / 1048576.0);
    return 0;
}

NRDHIP_API int nrdhip_library_desc(uint32_t out[5]) {
    out[0] = NRD_VERSION_MAJOR;
    out[1] = NRD_VERSION_MINOR;
    out[2] = NRD_VERSION_BUILD;
    out[3] = NRD_NORMAL_ENCODING;
    out[4] = NRD_ROUGHNESS_ENCODING;
    return 0;
}

NRDHIP_API const char* nrdhip_denoiser_string(uint32_t denoiser) {
    static const char* names[] = {"REBLUR_DIFFUSE", "REBLUR_DIFFUSE_OCCLUSION", "REBLUR_DIFFUSE_SH", "REBLUR_SPECULAR", "REBLUR_SPECULAR_OCCLUSION",
                                  "REBLUR_SPECULAR_SH", "REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION", "REBLUR_DIFFUSE_SPECULAR_SH",
                                  "REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION", "RELAX_DIFFUSE", "RELAX_DIFFUSE_SH", "RELAX_SPECULAR", "RELAX_SPECULAR_SH",
                                  "RELAX_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW", "SIGMA_SHADOW_TRANSLUCENCY", "REFERENCE"};
    return denoiser < (uint32_t)nrd::Denoiser::MAX_NUM ? names[denoiser] : "UNKNOWN";
}

NRDHIP_API uint32_t nrdhip_sizeof(uint32_t which) {
    switch (which) {
        case 0: return sizeof(nrd::CommonSettings);
        case 1: return sizeof(nrd::ReblurSettings);
        case 2: return sizeof(nrd::RelaxSettings);
        case 3: return sizeof(nrd::SigmaSettings);
        case 4: return sizeof(nrd::ReferenceSettings);
        case 5: return sizeof(nrdhip_create_desc);
        case 6: return sizeof(nrdhip_plane_info);
        case 7: return sizeof(nrdhip_dispatch_info);
        case 8: return sizeof(nrdhip_confidence_blur_desc);
        case 9: return sizeof(nrdhip_unpack_desc);
        case 10: return sizeof(nrdhip_taa_desc);
        case 11: return sizeof(nrdhip_frontend_pack_desc);
        case 12: return sizeof(nrdhip_compose_desc);
        case 13: return sizeof(nrdhip_transport);
    }
    return 0;
}

NRDHIP_API const char* nrdhip_last_error(nrdhip_instance* inst) { return inst ? inst->error.c_str() : g_createError.c_str(); }
}
