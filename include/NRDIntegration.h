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

// == nrd::IntegrationCreationDesc (Source/NRDSample.cpp:928-936). Fields that configure descriptor pools, in-flight frame
// rings and queue idling in the reference are ACCEPTED AND IGNORED here - a HIP stream orders everything and there are no
// descriptors: `name`, `queuedFrameNum`, `enableWholeLifetimeDescriptorCaching`, `autoWaitForIdle` (and
// ResourceSnapshot::restoreInitialState: HIP has no resource states to restore). The two format-conversion switches are
// refused (Recreate returns UNSUPPORTED): plane formats are part of the numerics contract.
struct IntegrationCreationDesc {
    char name[32] = "";
    uint16_t resourceWidth = 0;
    uint16_t resourceHeight = 0;
    uint8_t queuedFrameNum = 3;                        // ignored
    bool enableWholeLifetimeDescriptorCaching = false; // ignored
    bool promoteFloat16to32 = false;                   // not supported
    bool demoteFloat32to16 = false;                    // not supported
    bool autoWaitForIdle = true;                       // ignored
    bool submitAsGraph = false; // (extension) Denoise replays one HIP graph per frame instead of launching the passes one by one (NRDHIP_FLAG_GRAPH)
};

class Integration {
public:
    inline Integration() = default;
    inline ~Integration() { Destroy(); }
    Integration(const Integration&) = delete;
    Integration& operator=(const Integration&) = delete;

    // `device`: HIP device ordinal (the sample passes its nri::Device*): the pools are allocated there, and every later call makes it
    // current for its duration (and restores the caller's device), so a multi-GPU host can drive several Integrations from one thread.
    // Default -1 = no device of its own: every call runs on whatever device is current (a caller that never mentions a device keeps
    // its pools, its bound planes and its stream on one GPU without thinking about it)
    inline Result Recreate(const IntegrationCreationDesc& integrationDesc, const InstanceCreationDesc& instanceDesc, int device = -1) {
        return RecreateBand(integrationDesc, instanceDesc, device, nullptr);
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

    // upstream signature: (Identifier, const void*). Which settings struct the pointer refers to follows from the denoiser behind
    // `identifier` (nrdhip_denoiser_kind), exactly as in the reference's untyped call
    inline Result SetDenoiserSettings(Identifier identifier, const void* denoiserSettings) {
        if (!m_Instance)
            return Result::FAILURE;
        uint32_t kind = 0;
        if (nrdhip_denoiser_kind((nrdhip_instance*)m_Instance, identifier, &kind) != 0)
            return Result::INVALID_ARGUMENT;
        const size_t sizes[] = {sizeof(ReblurSettings), sizeof(RelaxSettings), sizeof(SigmaSettings), sizeof(ReferenceSettings)};
        return kind < 4 ? nrd::SetDenoiserSettings(*m_Instance, identifier, denoiserSettings, sizes[kind]) : Result::INVALID_ARGUMENT;
    }

    // `stream` plays the role of the sample's nri::CommandBuffer: all passes are enqueued on it, in order
    inline Result Denoise(const Identifier* denoisers, uint32_t denoisersNum, void* stream, ResourceSnapshot& resourceSnapshot) {
        if (!m_Instance)
            return Result::FAILURE;
        Result b = Bind(resourceSnapshot);
        if (b != Result::SUCCESS)
            return b;
        // final "states" are reported back like the reference does (:524-530); on HIP they are unchanged
        return (Result)nrdhip_denoise((nrdhip_instance*)m_Instance, denoisers, denoisersNum, stream);
    }

    // a snapshot is complete: slots it does not name lose last call's pointers
    inline Result Bind(ResourceSnapshot& resourceSnapshot) {
        nrdhip_unbind_all((nrdhip_instance*)m_Instance);
        for (size_t i = 0; i < (size_t)ResourceType::TRANSIENT_POOL; i++) {
            if (!resourceSnapshot.bound[i])
                continue;
            const Resource& r = resourceSnapshot.slots[i];
            int e = nrdhip_bind((nrdhip_instance*)m_Instance, (uint32_t)i, r.hip.ptr, r.hip.pitchBytes, (uint32_t)r.hip.format, r.hip.width, r.hip.height);
            if (e)
                return (Result)e;
        }
        return Result::SUCCESS;
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

protected:
    inline Result RecreateBand(const IntegrationCreationDesc& integrationDesc, const InstanceCreationDesc& instanceDesc, int device, const int32_t* band) {
        Destroy();
        if (integrationDesc.promoteFloat16to32 || integrationDesc.demoteFloat32to16)
            return Result::UNSUPPORTED;
        m_Desc = integrationDesc;
        m_Device = device;
        Result r = CreateInstance(instanceDesc, integrationDesc.resourceWidth, integrationDesc.resourceHeight, m_Instance, integrationDesc.submitAsGraph ? NRDHIP_FLAG_GRAPH : 0u, device, band);
        if (r != Result::SUCCESS)
            m_Instance = nullptr;
        return r;
    }

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

// One rank of a frame row-tiled across the GPUs of a node (BASELINE.json config 5; no counterpart in the reference, whose sample
// runs on a single adapter: Source/NRDSample.cpp:755-778). Same calls as Integration; Recreate additionally takes the band this
// rank owns, Denoise exchanges the halo rows between the passes (nrdhip_tiler_*: RCCL send / recv groups on a side stream, or a
// caller-supplied transport). `integrationDesc.resourceHeight` is the height of the WHOLE frame; planes bound to this rank hold
// rows [Row0(), Row0() + LocalHeight()) of it.
class TiledIntegration : public Integration {
public:
    inline ~TiledIntegration() { DestroyTiler(); }

    // First owned row of every rank plus frameHeight (world + 1 entries). Bands are whole TILE ROWS (16 pixel rows) and every rank gets
    // its share of them, not a remainder: 270 tile rows over 8 ranks are 34, 34, 34, 34, 34, 34, 33, 33. `tileRowCost` (optional, one
    // value per tile row, e.g. tiles with geometry + 0.15 x tiles without): the boundaries then balance the COST of the bands instead of
    // their height; every band keeps at least `minRows` rows (pass the halo: a neighbour's halo must come from one band). Boundaries are
    // fixed for the life of the instance (plane sizes depend on them): re-balancing = Recreate = an accumulation restart.
    static inline bool BandBounds(uint16_t frameHeight, int world, int32_t* bounds, const float* tileRowCost = nullptr, uint32_t minRows = 16) {
        const int n = (frameHeight + 15) / 16, minT = (int)((minRows + 15) / 16) > 1 ? (int)((minRows + 15) / 16) : 1;
        if (world <= 1) {
            bounds[0] = 0;
            bounds[1] = frameHeight;
            return true;
        }
        if (n < world * minT)
            return false;
        double total = 0.0;
        if (tileRowCost)
            for (int i = 0; i < n; i++)
                total += tileRowCost[i] > 0.0f ? (double)tileRowCost[i] : 0.0;
        int cut = 0;
        bounds[0] = 0;
        if (!tileRowCost || total <= 0.0) {
            for (int r = 0; r < world; r++) {
                cut += n / world + (r < n % world ? 1 : 0);
                bounds[r + 1] = cut * 16 > frameHeight ? frameHeight : cut * 16;
            }
            return true;
        }
        double cum = 0.0; // cost of tile rows [0, i)
        int i = 0;
        for (int k = 1; k < world; k++) {
            const int lo = cut + minT, hi = n - (world - k) * minT;
            const double target = total * k / world;
            while (i < lo)
                cum += tileRowCost[i] > 0.0f ? (double)tileRowCost[i] : 0.0, i++;
            double prev = cum;
            while (i < hi && cum < target)
                prev = cum, cum += tileRowCost[i] > 0.0f ? (double)tileRowCost[i] : 0.0, i++;
            if (i > lo && cum >= target && (target - prev) < (cum - target)) // the nearer of the two candidate cuts
                i--, cum = prev;
            cut = i;
            bounds[k] = cut * 16;
        }
        bounds[world] = frameHeight;
        return true;
    }
    // band of `rank` inside the bounds above: band = {frame height, first stored row, first owned row (local), owned rows}. false: the
    // frame cannot be cut that way - fewer tile rows than `world` bands of at least `haloRows` rows (BandBounds), more than 64 ranks, or
    // caller-supplied bounds that are not a monotone sequence of multiples of 16 from 0 to the frame height
    static inline bool BandOf(uint16_t frameHeight, int world, int rank, uint32_t haloRows, int32_t band[4], uint16_t& localHeight,
                              const int32_t* bounds = nullptr) {
        int32_t even[66];
        if (world < 1 || world > 64 || rank < 0 || rank >= world)
            return false;
        if (!bounds) {
            if (!BandBounds(frameHeight, world, even, nullptr, haloRows > 16u ? haloRows : 16u))
                return false;
            bounds = even;
        } else {
            if (bounds[0] != 0 || bounds[world] != (int32_t)frameHeight)
                return false;
            for (int r = 0; r < world; r++)
                if (bounds[r + 1] <= bounds[r] || (r + 1 < world && bounds[r + 1] % 16 != 0))
                    return false;
        }
        int own0 = bounds[rank], own1 = bounds[rank + 1];
        int row0 = own0 - (int)haloRows < 0 ? 0 : own0 - (int)haloRows;
        int row1 = own1 + (int)haloRows > frameHeight ? frameHeight : own1 + (int)haloRows;
        band[0] = frameHeight;
        band[1] = row0;
        band[2] = own0 - row0;
        band[3] = own1 - own0;
        localHeight = (uint16_t)(row1 - row0);
        return true;
    }

    inline Result Recreate(const IntegrationCreationDesc& integrationDesc, const InstanceCreationDesc& instanceDesc, int device, int rank, int world,
                           uint32_t haloRows, const nrdhip_transport* transport = nullptr, const int32_t* bounds = nullptr) {
        DestroyTiler();
        if (world > 64 || rank < 0 || rank >= world)
            return Result::INVALID_ARGUMENT;
        if (!BandOf(integrationDesc.resourceHeight, world, rank, haloRows, m_Band, m_LocalHeight, bounds))
            return Result::INVALID_ARGUMENT; // (e.g. a 100-row frame over 8 ranks: fewer tile rows than bands)
        IntegrationCreationDesc local = integrationDesc;
        local.resourceHeight = m_LocalHeight;
        Result r = RecreateBand(local, instanceDesc, device, m_Band);
        if (r != Result::SUCCESS)
            return r;
        return (Result)nrdhip_tiler_create((nrdhip_instance*)GetInstance(), rank, world, transport, &m_Tiler);
    }
    // RCCL transport: rank 0 calls GetUniqueId and hands the 128 bytes to the other ranks; every rank then calls InitRccl with its
    // GPU current
    static inline Result GetUniqueId(void* out128) { return (Result)nrdhip_tiler_rccl_unique_id(out128); }
    inline Result InitRccl(const void* uniqueId128) { return m_Tiler ? (Result)nrdhip_tiler_rccl_init(m_Tiler, uniqueId128) : Result::FAILURE; }

    inline int32_t Row0() const { return m_Band[1]; }
    inline uint16_t LocalHeight() const { return m_LocalHeight; }
    inline int32_t OwnFirst() const { return m_Band[2]; }
    inline int32_t OwnRows() const { return m_Band[3]; }

    // externally produced inputs arrive per band: refresh their halo rows from the neighbours (slots already in the snapshot)
    inline Result ExchangeInputs(const ResourceType* slots, uint32_t slotsNum, void* stream, ResourceSnapshot& resourceSnapshot) {
        if (!m_Tiler)
            return Result::FAILURE;
        Result b = Bind(resourceSnapshot);
        if (b != Result::SUCCESS)
            return b;
        return (Result)nrdhip_tiler_exchange_inputs(m_Tiler, (const uint32_t*)slots, slotsNum, stream);
    }

    inline Result Denoise(const Identifier* denoisers, uint32_t denoisersNum, void* stream, ResourceSnapshot& resourceSnapshot) {
        if (!m_Tiler)
            return Result::FAILURE;
        Result b = Bind(resourceSnapshot);
        if (b != Result::SUCCESS)
            return b;
        return (Result)nrdhip_tiler_denoise(m_Tiler, denoisers, denoisersNum, stream);
    }
    inline Result Finish(void* stream) { return m_Tiler ? (Result)nrdhip_tiler_finish(m_Tiler, stream) : Result::FAILURE; }
    inline const char* GetTilerError() const { return nrdhip_tiler_last_error(m_Tiler); }
    inline nrdhip_tiler* GetTiler() { return m_Tiler; }

    inline void Destroy() {
        DestroyTiler();
        Integration::Destroy();
    }

private:
    inline void DestroyTiler() {
        if (m_Tiler)
            nrdhip_tiler_destroy(m_Tiler);
        m_Tiler = nullptr;
    }
    nrdhip_tiler* m_Tiler = nullptr;
    int32_t m_Band[4] = {};
    uint16_t m_LocalHeight = 0;
};

} // namespace nrd
