// NRDIntegration.h - nrd::Integration for HIP: the drop-in twin of the reference's NRD Integration layer
// (External/NRD/Integration/NRDIntegration.hpp, absent from the reference tree; used by the sample at
// Source/NRDSample.cpp:10, 416-531, 625, 924-984, 2873, 3878-3879, 4080, 4124, 4150, 4221-4222).
//
// Same member functions, argument meaning, call order and error behaviour as the sample relies on:
//   Recreate(IntegrationCreationDesc, InstanceCreationDesc, device) -> Result   (:982; "!= SUCCESS -> return false")
//   NewFrame() ; SetCommonSettings(CommonSettings) ; SetDenoiserSettings(Identifier, const void*)   (:3878-3879, :4080...)
//   Denoise(const Identifier*, uint32_t, commandBuffer, ResourceSnapshot&)      (:521)
//   RecreatePipelines() (:2873), Destroy() (:744), Get{Total,Persistent,Aliasable}MemoryUsageInMb() (:1038)
// What changes is only what a "device", a "command buffer" and a "texture" are: a HIP device ordinal, a hipStream_t, and a
// device pointer + row pitch + nrd::Format. Everything is enqueued on the stream; nothing blocks, nothing throws.
#pragma once

#include "NRD.h"

#include <cstring>

namespace nrd {

// == nrd::Resource (Source/NRDSample.cpp:416-438): `.state` and `.userArg` round-trip untouched (HIP has no layouts)
struct Resource {
    struct {
        void* ptr = nullptr;      // device pointer of texel (0,0)
        uint32_t pitchBytes = 0;  // row pitch
        Format format = Format::MAX_NUM;
        uint16_t width = 0, height = 0;
    } hip;
    uint64_t state = 0;
    void* userArg = nullptr;
};

// == nrd::ResourceSnapshot (Source/NRDSample.cpp:442-501, 524-530)
struct ResourceSnapshot {
    Resource slots[(size_t)ResourceType::MAX_NUM] = {};
    bool bound[(size_t)ResourceType::MAX_NUM] = {};
    Resource unique[(size_t)ResourceType::MAX_NUM] = {};
    size_t uniqueNum = 0;
    bool restoreInitialState = false;

    inline void SetResource(ResourceType type, const Resource& resource) {
        slots[(size_t)type] = resource;
        bound[(size_t)type] = true;
        for (size_t i = 0; i < uniqueNum; i++)
            if (unique[i].hip.ptr == resource.hip.ptr)
                return;
        unique[uniqueNum++] = resource;
    }
};

// == nrd::IntegrationCreationDesc (Source/NRDSample.cpp:928-936)
struct IntegrationCreationDesc {
    char name[32] = "";
    uint16_t resourceWidth = 0;
    uint16_t resourceHeight = 0;
    uint8_t queuedFrameNum = 3;
    bool enableWholeLifetimeDescriptorCaching = false; // accepted for source compatibility; no descriptors on HIP
    bool promoteFloat16to32 = false;                   // not supported (plane formats are part of the numerics contract)
    bool demoteFloat32to16 = false;
    bool autoWaitForIdle = true;
};

class Integration {
public:
    inline Integration() = default;
    inline ~Integration() { Destroy(); }
    Integration(const Integration&) = delete;
    Integration& operator=(const Integration&) = delete;

    // `device`: HIP device ordinal the instance's pools live on (the sample passes its nri::Device*)
    inline Result Recreate(const IntegrationCreationDesc& integrationDesc, const InstanceCreationDesc& instanceDesc, int device = 0) {
        Destroy();
        if (integrationDesc.promoteFloat16to32 || integrationDesc.demoteFloat32to16)
            return Result::UNSUPPORTED;
        m_Desc = integrationDesc;
        m_Device = device;
        Result r = CreateInstance(instanceDesc, integrationDesc.resourceWidth, integrationDesc.resourceHeight, m_Instance);
        if (r != Result::SUCCESS)
            m_Instance = nullptr;
        return r;
    }

    // kernels are compiled ahead of time for gfx950: nothing to reload (the sample's shader hot-reload hook, :2866-2874)
    inline Result RecreatePipelines() { return m_Instance ? Result::SUCCESS : Result::FAILURE; }

    inline void NewFrame() {
        if (m_Instance)
            nrdhip_new_frame((nrdhip_instance*)m_Instance);
        m_FrameIndex++;
    }

    inline Result SetCommonSettings(const CommonSettings& commonSettings) {
        return m_Instance ? nrd::SetCommonSettings(*m_Instance, commonSettings) : Result::FAILURE;
    }

    // upstream signature: (Identifier, const void*). The size of the struct is implied by the denoiser kind.
    inline Result SetDenoiserSettings(Identifier identifier, const void* denoiserSettings) {
        if (!m_Instance)
            return Result::FAILURE;
        // try the four settings structs; the library accepts exactly the one matching the denoiser behind `identifier`
        const size_t sizes[] = {sizeof(ReblurSettings), sizeof(RelaxSettings), sizeof(SigmaSettings), sizeof(ReferenceSettings)};
        for (size_t s : sizes)
            if (nrd::SetDenoiserSettings(*m_Instance, identifier, denoiserSettings, s) == Result::SUCCESS)
                return Result::SUCCESS;
        return Result::INVALID_ARGUMENT;
    }

    // `stream` plays the role of the sample's nri::CommandBuffer: all passes are enqueued on it, in order
    inline Result Denoise(const Identifier* denoisers, uint32_t denoisersNum, void* stream, ResourceSnapshot& resourceSnapshot) {
        if (!m_Instance)
            return Result::FAILURE;
        for (size_t i = 0; i < (size_t)ResourceType::TRANSIENT_POOL; i++) {
            if (!resourceSnapshot.bound[i])
                continue;
            const Resource& r = resourceSnapshot.slots[i];
            int e = nrdhip_bind((nrdhip_instance*)m_Instance, (uint32_t)i, r.hip.ptr, r.hip.pitchBytes, (uint32_t)r.hip.format, r.hip.width, r.hip.height);
            if (e)
                return (Result)e;
        }
        // final "states" are reported back like the reference does (:524-530); on HIP they are unchanged
        return (Result)nrdhip_denoise((nrdhip_instance*)m_Instance, denoisers, denoisersNum, stream);
    }

    inline void Destroy() {
        if (m_Instance)
            DestroyInstance(*m_Instance);
        m_Instance = nullptr;
    }

    inline double GetTotalMemoryUsageInMb() const { return Mem(0); }
    inline double GetPersistentMemoryUsageInMb() const { return Mem(1); }
    inline double GetAliasableMemoryUsageInMb() const { return Mem(2); }
    inline const char* GetLastError() const { return nrdhip_last_error((nrdhip_instance*)m_Instance); }
    inline Instance* GetInstance() { return m_Instance; }

private:
    inline double Mem(int i) const {
        float v[3] = {};
        if (m_Instance)
            nrdhip_get_memory_mb((nrdhip_instance*)m_Instance, v);
        return v[i];
    }
    Instance* m_Instance = nullptr;
    IntegrationCreationDesc m_Desc = {};
    int m_Device = 0;
    uint32_t m_FrameIndex = 0;
};

} // namespace nrd
