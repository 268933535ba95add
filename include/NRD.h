// NRD.h - the nrd:: core C++ API of the MI355X-native backend (header-only veneer over the C-ABI of include/nrdhip.h).
//
// This is the surface the reference sample compiles against through "#include <NRD.h>" (Source/NRDSample.cpp:9); the
// reference's own implementation lives in the absent External/NRD submodule. Entry points kept source-compatible with the
// sample's uses:
//   nrd::GetLibraryDesc() ............ Source/NRDSample.cpp:1159-1162, 2915-2932, 3869
//   nrd::GetDenoiserString() ......... Source/NRDSample.cpp:1019
//   nrd::GetMaxAccumulatedFrameNum() . Source/NRDSample.cpp:2167, 2175   (NRDSettings.h)
// plus the instance-level API that sits beneath nrd::Integration: CreateInstance / DestroyInstance /
// GetInstanceDesc / SetCommonSettings / SetDenoiserSettings / GetComputeDispatches. On this backend a "dispatch" is a HIP
// kernel launch; GetComputeDispatches describes them, nrd::Integration (NRDIntegration.h) enqueues them on a hipStream_t.
#pragma once

#include "NRDDescs.h"
#include "NRDSettings.h"
#include "nrdhip.h"

#include <vector>

namespace nrd {

struct Instance; // opaque (== nrdhip_instance)

inline const LibraryDesc* GetLibraryDesc() {
    // every enumerator of nrd::Denoiser (Source/NRDSample.cpp:49-51, :871-921)
    static const Denoiser supported[] = {Denoiser::REBLUR_DIFFUSE, Denoiser::REBLUR_DIFFUSE_OCCLUSION, Denoiser::REBLUR_DIFFUSE_SH,
                                         Denoiser::REBLUR_SPECULAR, Denoiser::REBLUR_SPECULAR_OCCLUSION, Denoiser::REBLUR_SPECULAR_SH,
                                         Denoiser::REBLUR_DIFFUSE_SPECULAR, Denoiser::REBLUR_DIFFUSE_SPECULAR_OCCLUSION, Denoiser::REBLUR_DIFFUSE_SPECULAR_SH,
                                         Denoiser::REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION, Denoiser::RELAX_DIFFUSE, Denoiser::RELAX_DIFFUSE_SH, Denoiser::RELAX_SPECULAR, Denoiser::RELAX_SPECULAR_SH,
                                         Denoiser::RELAX_DIFFUSE_SPECULAR, Denoiser::RELAX_DIFFUSE_SPECULAR_SH,
                                         Denoiser::SIGMA_SHADOW, Denoiser::SIGMA_SHADOW_TRANSLUCENCY, Denoiser::REFERENCE};
    static LibraryDesc desc = {};
    uint32_t v[5] = {};
    nrdhip_library_desc(v);
    desc.supportedDenoisers = supported;
    desc.supportedDenoisersNum = (uint32_t)(sizeof(supported) / sizeof(supported[0]));
    desc.versionMajor = (uint8_t)v[0];
    desc.versionMinor = (uint8_t)v[1];
    desc.versionBuild = (uint8_t)v[2];
    desc.normalEncoding = (NormalEncoding)v[3];
    desc.roughnessEncoding = (RoughnessEncoding)v[4];
    return &desc;
}

inline const char* GetDenoiserString(Denoiser denoiser) { return nrdhip_denoiser_string((uint32_t)denoiser); }

// `device`: HIP device ordinal the pools are allocated on and the passes run on (-1 = whatever device is current at each call);
// `band`: row-band fields of nrdhip_create_desc for a row-tiled frame (nullptr = whole frame): {frame height, global row at
// local row 0, first owned local row, owned rows}
inline Result CreateInstance(const InstanceCreationDesc& desc, uint16_t resourceWidth, uint16_t resourceHeight, Instance*& instance, uint32_t flags = 0,
                             int device = -1, const int32_t* band = nullptr) {
    std::vector<nrdhip_denoiser_desc> dd(desc.denoisersNum);
    for (uint32_t i = 0; i < desc.denoisersNum; i++)
        dd[i] = {desc.denoisers[i].identifier, (uint32_t)desc.denoisers[i].denoiser};
    nrdhip_create_desc cd = {};
    cd.denoisers = dd.data();
    cd.denoisers_num = desc.denoisersNum;
    cd.resource_width = resourceWidth;
    cd.resource_height = resourceHeight;
    cd.flags = flags;
    cd.device_plus1 = device >= 0 ? (uint16_t)(device + 1) : (uint16_t)0;
    if (band) {
        cd.frame_height = (uint16_t)band[0];
        cd.band_row0 = band[1];
        cd.band_own_first = (uint16_t)band[2];
        cd.band_own_rows = (uint16_t)band[3];
    }
    nrdhip_instance* h = nullptr;
    int r = nrdhip_create(&cd, &h);
    instance = (Instance*)h;
    return (Result)r;
}

inline void DestroyInstance(Instance& instance) { nrdhip_destroy((nrdhip_instance*)&instance); }

inline Result SetCommonSettings(Instance& instance, const CommonSettings& commonSettings) {
    return (Result)nrdhip_set_common((nrdhip_instance*)&instance, &commonSettings, sizeof(commonSettings));
}

// denoiserSettings must point to the settings struct matching the denoiser behind `identifier`
// (ReblurSettings / RelaxSettings / SigmaSettings / ReferenceSettings), exactly like upstream's "const void*"
inline Result SetDenoiserSettings(Instance& instance, Identifier identifier, const void* denoiserSettings, size_t size) {
    return (Result)nrdhip_set_denoiser((nrdhip_instance*)&instance, identifier, denoiserSettings, size);
}

struct DispatchInfo {
    const char* name;
    const char* kernel;
    Identifier identifier;
    uint16_t gridWidth, gridHeight; // in 16x16 workgroups
    float algorithmicBytesPerPixel;
};

// Describes the kernel launches the given denoisers record for the current settings (upstream: GetComputeDispatches)
inline Result GetComputeDispatches(Instance& instance, const Identifier* identifiers, uint32_t identifiersNum, std::vector<DispatchInfo>& out) {
    uint32_t n = 0;
    int r = nrdhip_dispatch_count((nrdhip_instance*)&instance, identifiers, identifiersNum, &n);
    if (r)
        return (Result)r;
    out.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        nrdhip_dispatch_info di;
        r = nrdhip_dispatch_info_get((nrdhip_instance*)&instance, identifiers, identifiersNum, i, &di);
        if (r)
            return (Result)r;
        out[i] = {di.name, di.kernel, di.identifier, di.grid_width, di.grid_height, di.algorithmic_bytes_per_pixel};
    }
    return Result::SUCCESS;
}

} // namespace nrd
