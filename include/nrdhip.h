/* nrdhip.h - the C-ABI drop-in boundary of the MI355X-native NRD dispatch backend.
 *
 * Plain C: opaque handle, plain pointers and sizes, no C++/torch types. Every entry point cites the
 * reference interface it stands in for (paths relative to /root/reference). The reference's own NRD
 * library + NRD Integration layer (External/NRD, an empty submodule in the reference tree) sit exactly
 * here: between Sample::Denoise()/Sample::RenderFrame() and the GPU passes.
 *
 * Return convention: 0 = nrd::Result::SUCCESS, otherwise the nrd::Result code (never throws, never aborts;
 * the sample only tests "!= SUCCESS", Source/NRDSample.cpp:958-959, 982-983).
 *
 * Threading: like the reference (single render thread, single queue, Source/NRDSample.cpp:778, 3878-4154)
 * an instance is not thread-safe; all work is enqueued on the hipStream_t handed to nrdhip_denoise*.
 */
#ifndef NRDHIP_H
#define NRDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#    define NRDHIP_API __attribute__((visibility("default")))
#else
#    define NRDHIP_API
#endif

typedef struct nrdhip_instance nrdhip_instance;

/* == nrd::DenoiserDesc (Source/NRDSample.cpp:871-922): {identifier, nrd::Denoiser value} */
typedef struct nrdhip_denoiser_desc {
    uint32_t identifier;
    uint32_t denoiser;
} nrdhip_denoiser_desc;

enum {
    NRDHIP_FLAG_EXTERNAL_POOLS = 1u /* caller allocates pool planes (nrdhip_pool_info + nrdhip_bind_pool) */
};

/* == nrd::InstanceCreationDesc + nrd::IntegrationCreationDesc (Source/NRDSample.cpp:924-936).
 * Row-band fields describe the slice of a larger frame this instance owns when a frame is row-tiled
 * across GPUs; a whole-frame instance leaves them 0. */
typedef struct nrdhip_create_desc {
    const nrdhip_denoiser_desc* denoisers;
    uint32_t denoisers_num;
    uint16_t resource_width;  /* width of every full-resolution plane the instance sees */
    uint16_t resource_height; /* LOCAL height: band rows + halos when row-tiled */
    uint16_t frame_height;    /* 0 = resource_height; else the height of the whole (global) frame */
    uint16_t band_own_first;  /* first LOCAL row this instance must produce */
    uint16_t band_own_rows;   /* 0 = all rows */
    uint16_t reserved;
    int32_t band_row0;        /* global row stored at local row 0 (negative when the top halo is clipped) */
    uint32_t flags;
} nrdhip_create_desc;

typedef struct nrdhip_plane_info {
    void* ptr;            /* device pointer (NULL until bound/allocated) */
    uint32_t pitch_bytes; /* row pitch */
    uint32_t format;      /* nrd::Format */
    uint16_t width, height;
    uint32_t bytes_per_texel;
    const char* name;
} nrdhip_plane_info;

typedef struct nrdhip_dispatch_info {
    const char* name;      /* pass name, e.g. "REBLUR::TemporalAccumulation" */
    const char* kernel;    /* HIP kernel symbol */
    uint32_t identifier;   /* denoiser identifier the dispatch belongs to */
    uint16_t grid_width;   /* workgroups */
    uint16_t grid_height;
    uint16_t halo_rows;    /* rows beyond the owned band this pass may READ (row-tiling) */
    uint16_t written_num;  /* planes written (pool planes and output slots) */
    uint32_t written[12];  /* encoded: (pool << 16) | index, pool 0 = permanent, 1 = transient, 2 = slot(ResourceType) */
    uint32_t read_num;
    uint32_t read[24];
    float algorithmic_bytes_per_pixel; /* SURVEY.md 8d accounting rule applied to this pass */
} nrdhip_dispatch_info;

/* nrd::Integration::Recreate (Source/NRDSample.cpp:982): create the instance, size its pools. */
NRDHIP_API int nrdhip_create(const nrdhip_create_desc* desc, nrdhip_instance** out);
/* nrd::Integration::Destroy (Source/NRDSample.cpp:744) */
NRDHIP_API void nrdhip_destroy(nrdhip_instance* inst);
/* nrd::Integration::NewFrame (Source/NRDSample.cpp:3878) */
NRDHIP_API int nrdhip_new_frame(nrdhip_instance* inst);
/* nrd::Integration::SetCommonSettings (Source/NRDSample.cpp:3879, 4221); `settings` = nrd::CommonSettings */
NRDHIP_API int nrdhip_set_common(nrdhip_instance* inst, const void* settings, size_t size);
/* nrd::Integration::SetDenoiserSettings (Source/NRDSample.cpp:4080, 4124, 4150, 4222);
 * `settings` = nrd::ReblurSettings | RelaxSettings | SigmaSettings | ReferenceSettings by denoiser kind */
NRDHIP_API int nrdhip_set_denoiser(nrdhip_instance* inst, uint32_t identifier, const void* settings, size_t size);
/* nrd::ResourceSnapshot::SetResource (Source/NRDSample.cpp:447-501): bind a caller-owned plane to a slot.
 * `slot` = nrd::ResourceType, `format` = nrd::Format, device pointer + row pitch instead of nri::Texture*. */
NRDHIP_API int nrdhip_bind(nrdhip_instance* inst, uint32_t slot, void* dev_ptr, uint32_t pitch_bytes,
                           uint32_t format, uint16_t width, uint16_t height);
/* nrd::Integration::Denoise (Source/NRDSample.cpp:521): enqueue every pass of the given denoisers on `stream`. */
NRDHIP_API int nrdhip_denoise(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, void* hip_stream);

/* nrd::GetComputeDispatches equivalent (the core API beneath the Integration layer):
 * number of dispatches the given denoisers record this frame, their description, and ranged submission
 * (a row-tiling host exchanges halo rows between dispatches). */
NRDHIP_API int nrdhip_dispatch_count(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, uint32_t* count);
NRDHIP_API int nrdhip_dispatch_info_get(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, uint32_t index,
                                        nrdhip_dispatch_info* out);
NRDHIP_API int nrdhip_denoise_range(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, uint32_t first,
                                    uint32_t count, void* hip_stream);

/* nrd::GetInstanceDesc pools (permanent = 0, transient = 1): description and (external pools) binding */
NRDHIP_API int nrdhip_pool_size(nrdhip_instance* inst, uint32_t pool, uint32_t* count);
NRDHIP_API int nrdhip_pool_info(nrdhip_instance* inst, uint32_t pool, uint32_t index, nrdhip_plane_info* out);
NRDHIP_API int nrdhip_bind_pool(nrdhip_instance* inst, uint32_t pool, uint32_t index, void* dev_ptr, uint32_t pitch_bytes);

/* nrd::Integration::Get{Total,Persistent,Aliasable}MemoryUsageInMb (Source/NRDSample.cpp:1038): out[0..2] */
NRDHIP_API int nrdhip_get_memory_mb(nrdhip_instance* inst, float out[3]);

/* nrd::GetLibraryDesc (Source/NRDSample.cpp:1159-1162): out[0..4] = major, minor, build, normalEncoding, roughnessEncoding */
NRDHIP_API int nrdhip_library_desc(uint32_t out[5]);
/* nrd::GetDenoiserString (Source/NRDSample.cpp:1019) */
NRDHIP_API const char* nrdhip_denoiser_string(uint32_t denoiser);
/* sizeof() of the ABI structs as compiled, for binding self-checks:
 * 0 CommonSettings, 1 ReblurSettings, 2 RelaxSettings, 3 SigmaSettings, 4 ReferenceSettings,
 * 5 nrdhip_create_desc, 6 nrdhip_plane_info, 7 nrdhip_dispatch_info */
NRDHIP_API uint32_t nrdhip_sizeof(uint32_t which);
/* last error text of the instance (or of creation when inst == NULL) */
NRDHIP_API const char* nrdhip_last_error(nrdhip_instance* inst);

#ifdef __cplusplus
}
#endif
#endif /* NRDHIP_H */
