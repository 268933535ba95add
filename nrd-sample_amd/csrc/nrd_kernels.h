// nrd_kernels.h - kernel parameter blocks and launch entry points shared by the host dispatch core (nrdhip.cpp)
// and the HIP kernels (*.hip). One launch_* per pass of the pass graph.
#pragma once

#include "nrd_device.h"

namespace nrdhip {

#define NRDHIP_ROUGH_LUT_FLOATS (4096 + 2048)
struct ReblurParams {
    FrameConsts c;
    // nrd::ReblurSettings (Source/NRDSample.cpp:563-585 defaults, :4090-4124 per frame)
    float hp[4];
    float hitFactorDiff; // reblur_hitdist_factor(hp, 1): the diffuse signal's factor is uniform
    float planeDistanceSensitivity, lobeAngleFraction, roughnessFraction, minHitDistanceWeight;
    float minBlurRadius, maxBlurRadius, diffusePrepassBlurRadius, specularPrepassBlurRadius;
    float fastHistoryClampingSigmaScale, antilagSigmaScale, antilagSensitivity;
    float responsiveRoughnessThreshold, responsiveMinAccum;
    float maxA, maxFastA, maxStab;
    float maxASpec, maxFastASpec; // == maxA / maxFastA for REBLUR; RELAX has per-signal history caps
    int occlusion;                // 1: OCCLUSION variants - IN/OUT_*_HITDIST planes hold the normalised hit distance only
    int ioF16;                    //    ... as R16_SFLOAT (1) or R16_UNORM (0)
    int sh;                       // 1: SH variants - every signal carries a second texel (SH1) filtered with the weights of SH0
    int relax;                    // 1: RELAX front half (linear RGB + world-space hitT inputs, luma-moment history, HistoryFix writes History)
    int historyFixFrameNum, historyFixStride;
    // RELAX only (nrd::RelaxSettings / RelaxAntilagSettings, Source/NRDSample.cpp:1600-1606, :1626)
    float hfNormalPower;                           // HistoryFix normal weight = pow(N.Ns, historyFixEdgeStoppingNormalPower)
    float alAccel, alSpatial, alTemporal, alReset; // antilag: acceleration / spatial + temporal sigma scales / reset amount
    int reachPre, reachBlur, reachPost; // hard per-pass bound (pixels) on tap distance = halo rows of the pass
    float tapsPre[8][2], tapsPost[8][2]; // Poisson disk rotated for this frame (PrePass / PostBlur rotate per frame)
    uint32_t minMatDiff, minMatSpec;
    int clampEnabled;
    int prepassTrackOnly; // usePrepassOnlyForSpecularMotionEstimation: the PrePass passes the specular signal through (hit distance tracking still filtered)
    int returnHistLen;    // returnHistoryLengthInsteadOfOcclusion (OCCLUSION variants): OUT_*_HITDIST = accumulated frames * invMaxA
    float invMaxA;
    int antiFirefly;    // HistoryFix: clamp the luma to the centre-less 5x5 moments of the incoming signal
    float fireflyScale; //   sigma scale of that clamp (ReblurSettings::fireflySuppressorMinRelativeScale)
    int hasDiff, hasSpec;
    // resource slots
    PlaneRef inZ, inNR, inMV, inDiff, inSpec, confD, confS, outDiff, outSpec;
    PlaneRef inMix; // IN_DISOCCLUSION_THRESHOLD_MIX (R8_UNORM), only with FrameConsts::mixAvail
    PlaneRef inDiff1, inSpec1, outDiff1, outSpec1; // SH mode: IN/OUT_*_SH1
    PlaneRef outValidation;                        // OUT_VALIDATION (RGBA8), only with CommonSettings::enableValidation
    // PrepareInputs (checkerboard resolve / hit distance reconstruction): reads the raw slots, writes the planes the PrePass
    // then sees as inDiff / inSpec (/ inDiff1 / inSpec1)
    PlaneRef rawDiff, rawSpec, rawDiff1, rawSpec1;
    int prepared, checker, phaseDiff, phaseSpec, reconRadius;
    int prepSh1; // PrepareInputs also writes the SH1 copies (checkerboarded SH inputs / DIRECTIONAL_OCCLUSION)
    int dirInSnorm, dirOutSnorm; // ... whose planes are RGBA16_SNORM (1) or RGBA16_SFLOAT (0)
    int dirOcc;  // REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION: one {direction * h, h} texel in / out (split by PrepareInputs, merged by TS)
    // pools
    PlaneRef guide, guidePrev, data1, data1Prev, data1Tmp, data2, hist, fast, fastPrev, stab, stabPrev, tiles, tmp1, tmp2, hitTrack;
    // REBLUR flavours without SH: tap texels of Blur / PostBlur (nrd_device.h), [0] diffuse, [1] specular.
    // tapA: HistoryFix -> Blur, tapB: Blur -> PostBlur
    int tapTex;
    PlaneRef tapA[2], tapB[2];
    // ClassifyTiles only: the tile flags once more in LAUNCH order (FrameConsts::tileFlags of the passes behind it) through the inverse of
    // the tile table (tile -> forward launch index); nullptr when those passes launch over another grid than ClassifyTiles
    uint8_t* tileFlagsOut;
    const uint32_t* tileInv;
    // The roughness-only terms of the specular kernel set-up as a table over the 10-bit roughness code of the guide (nrd_reblur.hip
    // rough_terms): 1024 x {dominant factor, magic curve, lobe half angle, hit distance factor} = 16 KB per denoiser, written by the first
    // workgroup of every ClassifyTiles launch with the very functions a pixel would evaluate, read back by the spatial passes with one
    // 16-byte load per pixel (nullptr: the denoiser has no specular signal). Behind it (floats 4096 ..): 1024 x {1 - 2^(-200 r^2), log2 r},
    // the roughness-only terms of TemporalAccumulation's specular accumulation limit. NRDHIP_ROUGH_LUT_FLOATS floats in all
    float* roughLut;
};

// one RELAX A-trous iteration (nrd_reblur.hip k_relax_atrous)
struct AtrousParams {
    FrameConsts c;
    float depthSens, histThreshold, specularVarianceBoost;
    float phi[2], minLw[2]; // [0] diffuse, [1] specular
    float lobeAngleFraction, roughnessFraction;
    uint32_t minMatDiff, minMatSpec;
    int roughnessEdgeStopping;
    float lumRelax, normRelax, roughRelax; // {luminance, normal, roughness}EdgeStoppingRelaxation, saturated
    float lobeSlack;                       // specularLobeAngleSlack in radians
    int confDriven;                        // confidenceDrivenRelaxationMultiplier > 0 and confidence inputs available
    float confMult, confLumRelax, confNormRelax;
    PlaneRef confD, confS;
    int it, last, hasDiff, hasSpec, sh;
    PlaneRef guide, data1, data2, hist, mom, in, out, inDiff, inSpec, outDiff, outSpec, inDiff1, inSpec1, outDiff1, outSpec1;
    PlaneRef tiles; // RELAX::Tiles (ClassifyTiles): 1 = the tile has no geometry
    const float* roughLut; // ReblurParams::roughLut of the denoiser (this frame's ClassifyTiles wrote it): the lobe half angle by roughness code
};

struct SigmaParams {
    FrameConsts c;
    float planeDistanceSensitivity, maxStab;
    int translucency, outBpt;
    PlaneRef inZ, inNR, inMV, inPen, inTransl, out;
    PlaneRef inMix; // IN_DISOCCLUSION_THRESHOLD_MIX (R8_UNORM), only with FrameConsts::mixAvail
    PlaneRef guide, guidePrev, hist, histPrev, tiles, tilesSmooth, shadow1, pen1, shadow2;
};

struct ReferenceParams {
    FrameConsts c;
    float weight; // 1 / (1 + min(frames, maxAccumulatedFrameNum)), 1 on restart
    int restart;
    PlaneRef in, out, hist;
};

// grid = XCD-swizzled 16x16 tiles over the owned rows (nrd_device.h xcd_tile). Every launcher exists per projection flavour
// (nrd_device.h NRD_ORTHO): nrdhip::persp::launch_* (nrd_reblur.hip, nrd_sigma.hip) and nrdhip::ortho::launch_* (*_ortho.hip)
#define NRD_LAUNCH_DECLS \
    void launch_reference_accumulate(const ReferenceParams& p, hipStream_t s); \
    void launch_reblur_classify_tiles(const ReblurParams& p, hipStream_t s); \
    void launch_reblur_prepare_inputs(const ReblurParams& p, hipStream_t s); \
    void launch_reblur_validation(const ReblurParams& p, hipStream_t s); \
    void launch_reblur_spatial(const ReblurParams& p, int variant, hipStream_t s); \
    void launch_reblur_blur_radiance(const ReblurParams& p, hipStream_t s); \
    void launch_reblur_temporal_accumulation(const ReblurParams& p, hipStream_t s); \
    void launch_reblur_prepass_temporal_accumulation(const ReblurParams& p, hipStream_t s); \
    void launch_reblur_history_fix(const ReblurParams& p, hipStream_t s); \
    void launch_reblur_temporal_stabilization(const ReblurParams& p, hipStream_t s); \
    void launch_relax_atrous(const AtrousParams& p, hipStream_t s); \
    void launch_sigma_classify_tiles(const SigmaParams& p, hipStream_t s); \
    void launch_sigma_smooth_tiles(const SigmaParams& p, hipStream_t s); \
    void launch_sigma_blur(const SigmaParams& p, int pass, hipStream_t s); \
    void launch_sigma_temporal_stabilization(const SigmaParams& p, hipStream_t s);
// launch_reblur_spatial variant: 0 PrePass, 1 Blur, 2 PostBlur; launch_sigma_blur pass: 0 Blur, 1 PostBlur
namespace persp {
NRD_LAUNCH_DECLS
}
namespace ortho {
NRD_LAUNCH_DECLS
}
#undef NRD_LAUNCH_DECLS
// the launcher of the frame's projection flavour
#define NRD_PICK(consts, fn) ((consts).ortho ? ::nrdhip::ortho::fn : ::nrdhip::persp::fn)

} // namespace nrdhip
