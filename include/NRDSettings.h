// NRDSettings.h - settings structs of the nrd:: API surface (MI355X-native NRD backend).
//
// Field names follow the reference sample's uses (the only in-tree evidence of the API):
//   CommonSettings ................ Source/NRDSample.cpp:3835-3876
//   ReblurSettings ................ Source/NRDSample.cpp:563-585, 1515-1582, 2183-2184, 4090-4098
//   RelaxSettings ................. Source/NRDSample.cpp:543-561, 1585-1663, 2186-2189
//   SigmaSettings ................. Source/NRDSample.cpp:4072-4076, 1684, 2175-2177
//   ReferenceSettings ............. Source/NRDSample.cpp:1665-1667
//   history caps / helpers ........ Source/NRDSample.cpp:38, 2167, 2175
// Default values are this build's frozen operating point (DESIGN.md "settings defaults").
#pragma once

#include <cstdint>

namespace nrd {

constexpr uint32_t REBLUR_MAX_HISTORY_FRAME_NUM = 63;
constexpr uint32_t RELAX_MAX_HISTORY_FRAME_NUM = 255;
constexpr uint32_t SIGMA_MAX_HISTORY_FRAME_NUM = 7;
constexpr uint32_t REFERENCE_MAX_HISTORY_FRAME_NUM = 4095;

constexpr float REBLUR_DEFAULT_ACCUMULATION_TIME = 0.5f;  // seconds (30 frames @ 60 FPS)
constexpr float RELAX_DEFAULT_ACCUMULATION_TIME = 0.5f;   // seconds
constexpr float SIGMA_DEFAULT_ACCUMULATION_TIME = 0.084f; // seconds (5 frames @ 60 FPS)

enum class CheckerboardMode : uint8_t {
    OFF,
    BLACK,
    WHITE,
    MAX_NUM
};

enum class AccumulationMode : uint8_t {
    CONTINUE,          // common mode
    RESTART,           // discard history, keep resources
    CLEAR_AND_RESTART, // discard history and clear resources (NRDSample.cpp:3864)
    MAX_NUM
};

enum class HitDistanceReconstructionMode : uint8_t {
    OFF,
    AREA_3X3,
    AREA_5X5,
    MAX_NUM
};

// All matrices are column-major float[16] (memcpy of the sample's float4x4, NRDSample.cpp:3836-3839):
// element (row r, column c) lives at [c * 4 + r]; clip = M * view (column vectors).
struct CommonSettings {
    float viewToClipMatrix[16] = {};
    float viewToClipMatrixPrev[16] = {};
    float worldToViewMatrix[16] = {};
    float worldToViewMatrixPrev[16] = {};
    float worldPrevToWorldMatrix[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

    // IN_MV.xyz * motionVectorScale = {uv delta, uv delta, viewZ delta}; z = 0 selects 2D motion
    float motionVectorScale[3] = {1.0f, 1.0f, 0.0f};
    float cameraJitter[2] = {};
    float cameraJitterPrev[2] = {};

    uint16_t resourceSize[2] = {};
    uint16_t resourceSizePrev[2] = {};
    uint16_t rectSize[2] = {};
    uint16_t rectSizePrev[2] = {};

    float viewZScale = 1.0f;
    float timeDeltaBetweenFrames = 0.0f;
    float denoisingRange = 500000.0f;
    float disocclusionThreshold = 0.01f;
    float disocclusionThresholdAlternate = 0.05f;
    float cameraAttachedReflectionMaterialID = 999.0f;
    float strandMaterialID = 999.0f;
    float strandThickness = 80e-6f;
    float splitScreen = 0.0f;

    uint16_t printfAt[2] = {9999, 9999};
    float debug = 0.0f;
    uint32_t rectOrigin[2] = {};
    uint32_t frameIndex = 0;

    AccumulationMode accumulationMode = AccumulationMode::CONTINUE;
    bool isMotionVectorInWorldSpace = false;
    bool isHistoryConfidenceAvailable = false;
    bool isDisocclusionThresholdMixAvailable = false;
    bool isBaseColorMetalnessAvailable = false;
    bool enableValidation = false;
};

// normHitDist = saturate(hitDist / ((A + |viewZ| * B) * lerp(1, C, exp2(D * roughness^2))))
struct ReblurHitDistanceParameters {
    float A = 3.0f;
    float B = 0.1f;
    float C = 20.0f;
    float D = -25.0f;
};

struct ReblurAntilagSettings {
    float luminanceSigmaScale = 4.0f; // [1; 5]
    float luminanceSensitivity = 3.0f; // [1; 5]
};

struct ResponsiveAccumulationSettings {
    float roughnessThreshold = 0.0f;
    uint32_t minAccumulatedFrameNum = 3;
};

struct ReblurSettings {
    ReblurHitDistanceParameters hitDistanceParameters = {};
    ReblurAntilagSettings antilagSettings = {};
    ResponsiveAccumulationSettings responsiveAccumulationSettings = {};

    uint32_t maxAccumulatedFrameNum = 30;     // [0; REBLUR_MAX_HISTORY_FRAME_NUM]
    uint32_t maxFastAccumulatedFrameNum = 6;  // [0; maxAccumulatedFrameNum)
    uint32_t maxStabilizedFrameNum = REBLUR_MAX_HISTORY_FRAME_NUM; // 0 disables temporal stabilization
    uint32_t historyFixFrameNum = 3;          // [0; 5]
    uint32_t historyFixBasePixelStride = 14;  // > 0

    float diffusePrepassBlurRadius = 30.0f;  // pixels, 0 disables
    float specularPrepassBlurRadius = 50.0f; // pixels, 0 disables
    float minHitDistanceWeight = 0.1f;       // (0; 0.2]
    float minBlurRadius = 1.0f;              // pixels
    float maxBlurRadius = 30.0f;             // pixels
    float lobeAngleFraction = 0.15f;         // (0; 1]
    float roughnessFraction = 0.15f;         // (0; 1]
    float planeDistanceSensitivity = 0.02f;  // fraction of the frustum size
    float specularProbabilityThresholdsForMvModification[2] = {0.5f, 0.9f};
    float fireflySuppressorMinRelativeScale = 2.0f;
    float fastHistoryClampingSigmaScale = 2.0f; // [1; 3]

    CheckerboardMode checkerboardMode = CheckerboardMode::OFF;
    HitDistanceReconstructionMode hitDistanceReconstructionMode = HitDistanceReconstructionMode::OFF;
    uint8_t minMaterialForDiffuse = 4;  // >= 4 disables material-aware filtering
    uint8_t minMaterialForSpecular = 4;
    bool enableAntiFirefly = false;
    bool usePrepassOnlyForSpecularMotionEstimation = false;
    bool returnHistoryLengthInsteadOfOcclusion = false;
};

struct RelaxAntilagSettings {
    float accelerationAmount = 0.3f; // [0; 1]
    float spatialSigmaScale = 4.5f;
    float temporalSigmaScale = 0.5f;
    float resetAmount = 0.5f; // [0; 1]
};

struct RelaxSettings {
    RelaxAntilagSettings antilagSettings = {};

    uint32_t diffuseMaxAccumulatedFrameNum = 30;
    uint32_t specularMaxAccumulatedFrameNum = 30;
    uint32_t diffuseMaxFastAccumulatedFrameNum = 6;
    uint32_t specularMaxFastAccumulatedFrameNum = 6;
    uint32_t historyFixFrameNum = 3;
    uint32_t historyFixBasePixelStride = 14;
    uint32_t spatialVarianceEstimationHistoryThreshold = 3; // [0; 10]
    uint32_t atrousIterationNum = 5;                        // [2; 8]

    float diffusePrepassBlurRadius = 30.0f;
    float specularPrepassBlurRadius = 50.0f;
    float historyFixEdgeStoppingNormalPower = 8.0f;
    float fastHistoryClampingSigmaScale = 2.0f;
    float diffusePhiLuminance = 2.0f;
    float specularPhiLuminance = 1.0f;
    float diffuseMinLuminanceWeight = 0.0f;
    float specularMinLuminanceWeight = 0.0f;
    float lobeAngleFraction = 0.5f;
    float roughnessFraction = 0.15f;
    float specularVarianceBoost = 0.0f;
    float specularLobeAngleSlack = 0.15f; // degrees
    float depthThreshold = 0.003f;
    float minHitDistanceWeight = 0.1f;
    float luminanceEdgeStoppingRelaxation = 0.5f;
    float normalEdgeStoppingRelaxation = 0.3f;
    float roughnessEdgeStoppingRelaxation = 1.0f;
    float confidenceDrivenRelaxationMultiplier = 0.0f;
    float confidenceDrivenLuminanceEdgeStoppingRelaxation = 0.0f;
    float confidenceDrivenNormalEdgeStoppingRelaxation = 0.0f;

    CheckerboardMode checkerboardMode = CheckerboardMode::OFF;
    HitDistanceReconstructionMode hitDistanceReconstructionMode = HitDistanceReconstructionMode::OFF;
    uint8_t minMaterialForDiffuse = 4;
    uint8_t minMaterialForSpecular = 4;
    bool enableAntiFirefly = false;
    bool enableRoughnessEdgeStopping = true;
};

struct SigmaSettings {
    float lightDirection[3] = {0.0f, 0.0f, 0.0f}; // unit vector toward the light (NRDSample.cpp:4072-4076)
    float planeDistanceSensitivity = 0.02f;
    uint32_t maxStabilizedFrameNum = 5; // [0; SIGMA_MAX_HISTORY_FRAME_NUM]
};

struct ReferenceSettings {
    uint32_t maxAccumulatedFrameNum = 1024; // [0; REFERENCE_MAX_HISTORY_FRAME_NUM]
};

// "Frames for a time window" helper (NRDSample.cpp:2167, 2175)
inline uint32_t GetMaxAccumulatedFrameNum(float accumulationTime, float fps) {
    return (uint32_t)(accumulationTime * fps + 0.5f);
}

} // namespace nrd
