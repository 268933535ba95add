// NRDDescs.h - MI355X-native NRD backend: enums and descriptors of the nrd:: API surface.
//
// Written from scratch against the *call sites* of the reference sample (the NRD headers
// themselves are an empty submodule in the reference tree):
//   Denoiser enumerators ......... Source/NRDSample.cpp:49-51, 871-921, 1016 (REFERENCE is last)
//   DenoiserDesc / Identifier ..... Source/NRDSample.cpp:226, 871-922
//   InstanceCreationDesc .......... Source/NRDSample.cpp:924-926
//   ResourceType slots ............ Source/NRDSample.cpp:447-501
//   Result ........................ Source/NRDSample.cpp:958, 982
//   LibraryDesc / NormalEncoding .. Source/NRDSample.cpp:1159-1162, 2915-2932
#pragma once

#include <cstdint>
#include <cstddef>

#define NRD_VERSION_MAJOR 4
#define NRD_VERSION_MINOR 17
#define NRD_VERSION_BUILD 0

// Build-time encodings the reference fixes through CMake (CMakeLists.txt:136-137)
#ifndef NRD_NORMAL_ENCODING
#    define NRD_NORMAL_ENCODING 2 // R10_G10_B10_A2_UNORM
#endif
#ifndef NRD_ROUGHNESS_ENCODING
#    define NRD_ROUGHNESS_ENCODING 1 // LINEAR
#endif

namespace nrd {

typedef uint32_t Identifier;

enum class Result : uint32_t {
    SUCCESS,
    FAILURE,
    INVALID_ARGUMENT,
    UNSUPPORTED,
    NON_UNIQUE_IDENTIFIER,
    MAX_NUM
};

// Order matters: the sample iterates "i <= (uint32_t)Denoiser::REFERENCE" (NRDSample.cpp:1016)
enum class Denoiser : uint32_t {
    REBLUR_DIFFUSE,
    REBLUR_DIFFUSE_OCCLUSION,
    REBLUR_DIFFUSE_SH,
    REBLUR_SPECULAR,
    REBLUR_SPECULAR_OCCLUSION,
    REBLUR_SPECULAR_SH,
    REBLUR_DIFFUSE_SPECULAR,
    REBLUR_DIFFUSE_SPECULAR_OCCLUSION,
    REBLUR_DIFFUSE_SPECULAR_SH,
    REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION,
    RELAX_DIFFUSE,
    RELAX_DIFFUSE_SH,
    RELAX_SPECULAR,
    RELAX_SPECULAR_SH,
    RELAX_DIFFUSE_SPECULAR,
    RELAX_DIFFUSE_SPECULAR_SH,
    SIGMA_SHADOW,
    SIGMA_SHADOW_TRANSLUCENCY,
    REFERENCE,
    MAX_NUM
};

// Resource slots (NRDSample.cpp:447-501 binds every one the sample uses)
enum class ResourceType : uint32_t {
    // Common guides
    IN_MV,               // RGBA16F: xy = screen motion (scaled by motionVectorScale.xy), z = viewZ delta (2.5D)
    IN_NORMAL_ROUGHNESS, // R10G10B10A2_UNORM (NRD_NORMAL_ENCODING 2): oct normal, roughness, materialID/3
    IN_VIEWZ,            // R32F linear view depth, |viewZ| > denoisingRange = sky
    IN_BASECOLOR_METALNESS,
    IN_DIFF_CONFIDENCE, // x = 0..1 history confidence, any resolution (sampled by uv)
    IN_SPEC_CONFIDENCE,
    IN_DISOCCLUSION_THRESHOLD_MIX,

    // Noisy inputs
    IN_DIFF_RADIANCE_HITDIST, // RGBA16F
    IN_SPEC_RADIANCE_HITDIST, // RGBA16F
    IN_DIFF_HITDIST,
    IN_SPEC_HITDIST,
    IN_DIFF_DIRECTION_HITDIST,
    IN_DIFF_SH0,
    IN_DIFF_SH1,
    IN_SPEC_SH0,
    IN_SPEC_SH1,
    IN_PENUMBRA,     // R16F
    IN_TRANSLUCENCY, // RGBA8
    IN_SIGNAL,       // RGBA16F

    // Outputs
    OUT_DIFF_RADIANCE_HITDIST,
    OUT_SPEC_RADIANCE_HITDIST,
    OUT_DIFF_SH0,
    OUT_DIFF_SH1,
    OUT_SPEC_SH0,
    OUT_SPEC_SH1,
    OUT_DIFF_HITDIST,
    OUT_SPEC_HITDIST,
    OUT_DIFF_DIRECTION_HITDIST,
    OUT_SHADOW_TRANSLUCENCY, // RGBA8 (R8 for SIGMA_SHADOW)
    OUT_SIGNAL,
    OUT_VALIDATION, // RGBA8

    // Pools (internal)
    TRANSIENT_POOL,
    PERMANENT_POOL,

    MAX_NUM
};

enum class Format : uint32_t {
    R8_UNORM,
    R8_UINT,
    RGBA8_UNORM,
    R16_UINT,
    R16_SFLOAT,
    RG16_SFLOAT,
    RGBA16_SFLOAT,
    R32_UINT,
    R32_SFLOAT,
    RG32_UINT, // 8-byte packed guide plane {viewZ, normal+roughness}
    RGBA32_SFLOAT,
    R10_G10_B10_A2_UNORM,
    RGBA32_UINT, // 16-byte packed diff+spec radiance plane (2 x RGBA16F)
    R16_UNORM,   // OCCLUSION variants: normalised hit distance (Source/NRDSample.cpp:2934-2937)
    RGBA16_SNORM, // DIRECTIONAL_OCCLUSION: {direction * hitDist, hitDist} (Source/NRDSample.cpp:2937)
    MAX_NUM
};

enum class NormalEncoding : uint8_t {
    RGBA8_UNORM,
    RGBA8_SNORM,
    R10_G10_B10_A2_UNORM,
    RGBA16_UNORM,
    RGBA16_SNORM,
    MAX_NUM
};

enum class RoughnessEncoding : uint8_t {
    SQ_LINEAR,
    LINEAR,
    SQRT_LINEAR,
    MAX_NUM
};

struct LibraryDesc {
    const Denoiser* supportedDenoisers;
    uint32_t supportedDenoisersNum;
    uint8_t versionMajor;
    uint8_t versionMinor;
    uint8_t versionBuild;
    NormalEncoding normalEncoding;
    RoughnessEncoding roughnessEncoding;
};

struct DenoiserDesc {
    Identifier identifier;
    Denoiser denoiser;
};

struct InstanceCreationDesc {
    const DenoiserDesc* denoisers;
    uint32_t denoisersNum;
};

// One plane of an internal pool: what the integration layer must allocate
struct TextureDesc {
    Format format;
    uint16_t downsampleFactor; // 1 = full resolution, 16 = one texel per 16x16 tile
};

struct ResourceDesc {
    ResourceType type;
    uint16_t indexInPool; // for TRANSIENT_POOL / PERMANENT_POOL
    bool isWritten;       // storage (true) or read-only (false) binding
};

// Formats this backend accepts per resource slot (the sample's texture formats, Source/NRDSample.cpp:2934-2990); a slot bound
// with anything else makes Denoise fail with INVALID_ARGUMENT instead of silently misreading memory
inline bool IsFormatAllowed(ResourceType type, Format format) {
    switch (type) {
        case ResourceType::IN_NORMAL_ROUGHNESS: return format == Format::R10_G10_B10_A2_UNORM;
        case ResourceType::IN_VIEWZ: return format == Format::R32_SFLOAT;
        case ResourceType::IN_DIFF_HITDIST:
        case ResourceType::IN_SPEC_HITDIST:
        case ResourceType::OUT_DIFF_HITDIST:
        case ResourceType::OUT_SPEC_HITDIST: return format == Format::R16_UNORM || format == Format::R16_SFLOAT;
        case ResourceType::IN_PENUMBRA: return format == Format::R16_SFLOAT;
        case ResourceType::IN_DISOCCLUSION_THRESHOLD_MIX: return format == Format::R8_UNORM;
        case ResourceType::IN_TRANSLUCENCY:
        case ResourceType::OUT_VALIDATION: return format == Format::RGBA8_UNORM;
        case ResourceType::OUT_SHADOW_TRANSLUCENCY: return format == Format::RGBA8_UNORM || format == Format::R8_UNORM;
        case ResourceType::IN_DIFF_DIRECTION_HITDIST:
        case ResourceType::OUT_DIFF_DIRECTION_HITDIST: return format == Format::RGBA16_SFLOAT || format == Format::RGBA16_SNORM; // Source/NRDSample.cpp:2937
        default: return format == Format::RGBA16_SFLOAT; // motion vectors, radiance / SH / direction planes, confidence, REFERENCE signal
    }
}

// One recorded kernel launch ("dispatch" in the reference's vocabulary)
struct DispatchDesc {
    const char* name;
    Identifier identifier;
    const ResourceDesc* resources;
    uint32_t resourcesNum;
    const uint8_t* constantBufferData;
    uint32_t constantBufferDataSize;
    uint16_t pipelineIndex; // index of the HIP kernel in InstanceDesc::pipelines
    uint16_t gridWidth;     // in workgroups
    uint16_t gridHeight;
};

struct PipelineDesc {
    const char* kernelName; // HIP kernel symbol
    uint16_t workgroupWidth;
    uint16_t workgroupHeight;
};

struct InstanceDesc {
    const PipelineDesc* pipelines;
    uint32_t pipelinesNum;
    const TextureDesc* permanentPool;
    uint32_t permanentPoolSize;
    const TextureDesc* transientPool;
    uint32_t transientPoolSize;
    uint32_t constantBufferMaxDataSize;
};

} // namespace nrd
