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
    NRDHIP_FLAG_EXTERNAL_POOLS = 1u, /* caller allocates pool planes (nrdhip_pool_info + nrdhip_bind_pool) */
    /* nrdhip_denoise submits the frame as ONE HIP graph launch: the dispatch list is stream-captured every frame (no GPU work, a few
     * microseconds per node), the executable graph the instance keeps for that identifier list takes the new kernel arguments through
     * hipGraphExecUpdate (it is only re-instantiated when the list changes shape: a CLEAR_AND_RESTART frame with its clears) and is launched
     * on the caller's stream. Needs a capturable stream: on the legacy default stream (NULL) the passes are launched one by one as
     * without the flag. Results are bit-identical either way; nrdhip_graph_stats reports what happened. */
    NRDHIP_FLAG_GRAPH = 2u,
    /* one dispatch per pass of the pass graph. By default the REBLUR radiance flavours (REBLUR_DIFFUSE / _SPECULAR / _DIFFUSE_SPECULAR)
     * run PrePass and TemporalAccumulation as ONE dispatch, "REBLUR::PrePassTemporalAccumulation": TemporalAccumulation reads the
     * PrePass result at its own pixel only, so it stays in registers and the Tmp1 plane is never touched. Outputs and every other
     * pool plane are bit-identical either way; the flag exists for callers that want the PrePass result as a plane (tests,
     * debugging). */
    NRDHIP_FLAG_SEPARATE_PASSES = 4u
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
    uint16_t device_plus1;    /* 0 = whatever HIP device is current at each call; d + 1 = pools and kernels on device d (made current
                               * for the duration of create / denoise / destroy and restored: nrd::Integration::Recreate(.., device)) */
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
    /* Row tiling, per read plane: how far beyond the rows it computes THIS dispatch reads read[i]. 0 = at the thread's own pixel only;
     * N = through a spatial footprint of up to N rows (halo_rows is the largest of them); NRDHIP_READ_REPROJECTED = previous-frame
     * state fetched at motion-displaced positions (the band's motion allowance + one footprint row). A plane a dispatch list does
     * not annotate reports halo_rows - the conservative answer. */
    uint16_t read_rows[24];
    /* NRDHIP_DISPATCH_ALL_ROWS: a pointwise pass over external inputs (the ClassifyTiles passes): it runs on EVERY row a band stores -
     * owned rows and halo rows alike, the inputs' halo rows being valid after nrdhip_tiler_exchange_inputs - so what it writes is
     * complete on every rank and is never exchanged */
    uint32_t flags;
    /* Row tiling, per written plane: written[i] holds 16-byte TAP TEXELS {guide texel | signal} whose first 8 bytes are a copy of the same
     * pixel's 8-byte texel in pool plane written_prefix[i] (the guide plane, which a NRDHIP_DISPATCH_ALL_ROWS pass completes on every
     * rank): only the last 8 bytes of each texel have to travel, the receiver puts the first 8 back from its own copy of that plane.
     * NRDHIP_NO_PLANE: the plane travels as it is. */
    uint32_t written_prefix[12];
} nrdhip_dispatch_info;
enum { NRDHIP_READ_REPROJECTED = 0xFFFFu };
enum { NRDHIP_DISPATCH_ALL_ROWS = 1u };
enum { NRDHIP_NO_PLANE = 0xFFFFFFFFu };

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
 * `slot` = nrd::ResourceType, `format` = nrd::Format, device pointer + row pitch instead of nri::Texture*.
 * `width` / `height` (texels) are mandatory: nrdhip_denoise checks every plane a pass touches against the rect (full rect width;
 * half of it for the noisy signal inputs of a checkerboarded denoiser; confidence planes any size) and refuses a smaller one. */
NRDHIP_API int nrdhip_bind(nrdhip_instance* inst, uint32_t slot, void* dev_ptr, uint32_t pitch_bytes,
                           uint32_t format, uint16_t width, uint16_t height);
/* forget every slot binding (a ResourceSnapshot is complete: slots absent from it must not keep last frame's pointers) */
NRDHIP_API int nrdhip_unbind_all(nrdhip_instance* inst);
/* nrd::Integration::Denoise (Source/NRDSample.cpp:521): enqueue every pass of the given denoisers on `stream`.
 * A denoiser's permanent planes are cleared (on `stream`) in front of its first pass on a CLEAR_AND_RESTART frame and on its very first
 * frame whatever the mode: the passes leave tiles without geometry unwritten, so caller-allocated pools need no initialisation. */
NRDHIP_API int nrdhip_denoise(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, void* hip_stream);

/* Checkpoint / resume of a denoiser's history ("dump / load of the permanent pool"; the reference keeps its history in GPU
 * textures only and can but reset it, Source/NRDSample.cpp:3864). Everything a denoiser carries from one frame to the next is its
 * permanent planes (nrdhip_pool_info gives pointer, pitch and size of each: copy them out / in) and these counters; read them between
 * frames, write them - with the planes restored - before the next frame's SetCommonSettings. A restored instance continues bit-identically. */
typedef struct nrdhip_history_state {
    uint32_t frame_counter;      /* frames denoised so far (its parity selects the ping-pong planes) */
    uint32_t frames_since_reset; /* REFERENCE: accumulated frame count */
    uint32_t history_valid;      /* 0 before the first frame */
    uint32_t reserved;
} nrdhip_history_state;
NRDHIP_API int nrdhip_get_history_state(nrdhip_instance* inst, uint32_t identifier, nrdhip_history_state* out);
NRDHIP_API int nrdhip_set_history_state(nrdhip_instance* inst, uint32_t identifier, const nrdhip_history_state* in);

/* NRDHIP_FLAG_GRAPH bookkeeping: {frames replayed through the graph, executable graphs instantiated, frames that fell back to direct
 * launches (default stream, capture unsupported)} */
NRDHIP_API int nrdhip_graph_stats(nrdhip_instance* inst, uint32_t out[3]);

/* nrd::GetComputeDispatches equivalent (the core API beneath the Integration layer):
 * number of dispatches the given denoisers record this frame, their description, and ranged submission
 * (a row-tiling host exchanges halo rows between dispatches). */
NRDHIP_API int nrdhip_dispatch_count(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, uint32_t* count);
NRDHIP_API int nrdhip_dispatch_info_get(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, uint32_t index,
                                        nrdhip_dispatch_info* out);
NRDHIP_API int nrdhip_denoise_range(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, uint32_t first,
                                    uint32_t count, void* hip_stream);

/* One dispatch restricted to LOCAL rows [row_first, row_first + row_count) of the rows this instance owns (clipped to them;
 * multiples of 16 keep whole tiles together). A row-tiling host launches the boundary strips of a pass first, starts the halo
 * exchange of those rows, and computes the interior while the exchange is in flight. `part`: bit 0 = first part of
 * this dispatch this frame (performs the CLEAR_AND_RESTART clear when due), bit 1 = last part (advances the frame state when the
 * dispatch is the denoiser's last). nrdhip_denoise_range(.., index, 1, ..) == nrdhip_denoise_rows(.., index, 0, all rows, 3, ..). */
enum { NRDHIP_PART_FIRST = 1u, NRDHIP_PART_LAST = 2u };
NRDHIP_API int nrdhip_denoise_rows(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, uint32_t index,
                                   uint32_t row_first, uint32_t row_count, uint32_t part, void* hip_stream);

/* nrd::GetInstanceDesc pools (permanent = 0, transient = 1): description and (external pools) binding */
NRDHIP_API int nrdhip_pool_size(nrdhip_instance* inst, uint32_t pool, uint32_t* count);
NRDHIP_API int nrdhip_pool_info(nrdhip_instance* inst, uint32_t pool, uint32_t index, nrdhip_plane_info* out);
NRDHIP_API int nrdhip_bind_pool(nrdhip_instance* inst, uint32_t pool, uint32_t index, void* dev_ptr, uint32_t pitch_bytes);

/* Introspection used by the layers above the C-ABI (the C++ nrd::Integration veneer, the row tiler):
 * nrdhip_denoiser_kind - which settings struct `identifier` takes: 0 ReblurSettings, 1 RelaxSettings, 2 SigmaSettings,
 *   3 ReferenceSettings (nrd::Integration::SetDenoiserSettings is one untyped call in the reference, Source/NRDSample.cpp:4080-4150);
 * nrdhip_get_band - {frame height, global row at local row 0, first owned local row, owned rows (resolved), local height};
 * nrdhip_slot_info - the plane currently bound to a resource slot (ptr NULL = unbound). */
NRDHIP_API int nrdhip_denoiser_kind(nrdhip_instance* inst, uint32_t identifier, uint32_t* kind);
NRDHIP_API int nrdhip_get_band(nrdhip_instance* inst, int32_t out[5]);
/* Row tiling: the LOCAL rows [first_local_row, first_local_row + rows) on which the previous frame's permanent planes are current on this
 * instance (rows = 0: every stored row - a whole-frame instance, the default). A row tiler refreshes only the rows of last frame's state
 * that reprojection may reach (owned rows +- the motion allowance of nrdhip_required_halo + 2), not the whole stored halo; with the
 * window set, a reprojection footprint (surface or virtual motion) that lands beyond it is REJECTED like one that leaves the frame - the
 * pixel restarts its history - instead of reading rows no exchange has written this frame. `motion_rows` of nrdhip_required_halo must
 * therefore bound the vertical displacement of BOTH the surface motion and the specular virtual motion for the tiled frame to stay
 * bit-identical to the single-GPU frame; beyond it the result is deterministic and self-consistent, but no longer identical.
 * No counterpart in the reference (single adapter). Both row tilers (nrdhip_tiler_*, nrd-sample_amd/tiler.py) set it from their plan. */
NRDHIP_API int nrdhip_set_history_rows(nrdhip_instance* inst, int32_t first_local_row, uint32_t rows);
/* HIP device ordinal the instance was created for (nrdhip_create_desc::device_plus1 - 1), -1 = "whatever is current at each call" */
NRDHIP_API int nrdhip_get_device(nrdhip_instance* inst);
NRDHIP_API int nrdhip_slot_info(nrdhip_instance* inst, uint32_t slot, nrdhip_plane_info* out);

/* nrd::Integration::Get{Total,Persistent,Aliasable}MemoryUsageInMb (Source/NRDSample.cpp:1038): out[0..2] */
NRDHIP_API int nrdhip_get_memory_mb(nrdhip_instance* inst, float out[3]);

/* nrd::GetLibraryDesc (Source/NRDSample.cpp:1159-1162): out[0..4] = major, minor, build, normalEncoding, roughnessEncoding */
NRDHIP_API int nrdhip_library_desc(uint32_t out[5]);
/* nrd::GetDenoiserString (Source/NRDSample.cpp:1019) */
NRDHIP_API const char* nrdhip_denoiser_string(uint32_t denoiser);
/* sizeof() of the ABI structs as compiled, for binding self-checks:
 * 0 CommonSettings, 1 ReblurSettings, 2 RelaxSettings, 3 SigmaSettings, 4 ReferenceSettings,
 * 5 nrdhip_create_desc, 6 nrdhip_plane_info, 7 nrdhip_dispatch_info, 8 nrdhip_confidence_blur_desc, 9 nrdhip_unpack_desc,
 * 10 nrdhip_taa_desc, 11 nrdhip_frontend_pack_desc, 12 nrdhip_compose_desc, 13 nrdhip_transport */
NRDHIP_API uint32_t nrdhip_sizeof(uint32_t which);
/* last error text of the instance (or of creation when inst == NULL) */
NRDHIP_API const char* nrdhip_last_error(nrdhip_instance* inst);

/* ===================================================================================================================
 * Row tiler: one frame row-tiled across the GPUs of a node (BASELINE.json config 5, SURVEY.md 8e scheme A). No counterpart in
 * the reference (single adapter, single queue: Source/NRDSample.cpp:755-778). One rank per GPU; every rank creates its
 * instance with the band fields of nrdhip_create_desc (owned rows + `halo` stored rows towards each neighbour, halo from
 * nrdhip_required_halo), then a tiler on top of it. nrdhip_tiler_denoise == nrdhip_denoise + the halo-row exchanges between the
 * dispatches; the result is bit-identical to the single-GPU frame, or the call fails (a pass that reads farther than the stored
 * halo is refused, never clamped). Rows travel point to point between row neighbours: RCCL (ncclSend / ncclRecv groups on a
 * side stream, overlapped with the interior of the dispatch) after nrdhip_tiler_rccl_init, or the caller's transport callbacks.
 * =================================================================================================================== */
typedef struct nrdhip_tiler nrdhip_tiler;
/* caller-supplied transport (tests, hosts with their own fabric layer): all calls of one exchange arrive between group_begin
 * and group_end (both optional), sends and receives toward one peer in matching order on both sides. `hip_stream` is the
 * stream the rows were produced on - a host transport synchronises it before reading. `dev_ptr` is device memory of this rank: rows of a
 * pool plane or of a bound slot, or a staging buffer of the tiler (the packed signal halves of tap-texel rows, nrdhip_dispatch_info::
 * written_prefix) - a transport must not assume it lies inside a plane the caller allocated. Return 0 on success. */
typedef struct nrdhip_transport {
    void* user;
    int (*group_begin)(void* user);
    int (*send)(void* user, const void* dev_ptr, size_t bytes, int peer_rank, void* hip_stream);
    int (*recv)(void* user, void* dev_ptr, size_t bytes, int peer_rank, void* hip_stream);
    int (*group_end)(void* user, void* hip_stream);
    /* NRDHIP_TRANSPORT_STREAM_ORDERED: send / recv only ENQUEUE asynchronous work on the stream they are handed and return at once, the
     * way ncclSend / ncclRecv do. The tiler then treats the transport exactly like RCCL: it hands it its side stream and orders that
     * stream against the compute stream with the same events (compute -> side stream before an exchange, side stream -> compute before
     * the next dispatch, a separate event for the deferred rows). Without the flag (0) the callbacks receive the compute stream and
     * are expected to have finished the transfer when they return (host transports: they synchronise the stream themselves). */
    uint32_t flags;
} nrdhip_transport;
enum { NRDHIP_TRANSPORT_STREAM_ORDERED = 1u };
/* rows a band must store beyond its owned rows: max read reach of the dispatches of `identifiers` with the settings currently
 * set + `motion_rows` (the largest vertical motion, in rows, reprojection may follow), rounded up to 16 */
NRDHIP_API int nrdhip_required_halo(nrdhip_instance* inst, const uint32_t* identifiers, uint32_t n, uint32_t motion_rows, uint32_t* out);
/* transport NULL = RCCL (then call nrdhip_tiler_rccl_init before the first exchange) */
NRDHIP_API int nrdhip_tiler_create(nrdhip_instance* inst, int rank, int world, const nrdhip_transport* transport, nrdhip_tiler** out);
NRDHIP_API void nrdhip_tiler_destroy(nrdhip_tiler* tiler);
/* ncclGetUniqueId on rank 0 (128 bytes; hand it to the other ranks by any means), ncclCommInitRank on every rank with the
 * rank's GPU current */
NRDHIP_API int nrdhip_tiler_rccl_unique_id(void* out128);
NRDHIP_API int nrdhip_tiler_rccl_init(nrdhip_tiler* tiler, const void* unique_id128);
/* refresh the halo rows of externally produced inputs (slots = nrd::ResourceType values already bound with nrdhip_bind) */
NRDHIP_API int nrdhip_tiler_exchange_inputs(nrdhip_tiler* tiler, const uint32_t* slots, uint32_t n, void* hip_stream);
NRDHIP_API int nrdhip_tiler_denoise(nrdhip_tiler* tiler, const uint32_t* identifiers, uint32_t n, void* hip_stream);
/* make `hip_stream` wait for rows still travelling (those only the next frame reads); call before reading pool planes / at exit */
NRDHIP_API int nrdhip_tiler_finish(nrdhip_tiler* tiler, void* hip_stream);
/* Bring-up on ONE GPU: one real exchange group of the RCCL transport - ncclGroupStart, ncclSend + ncclRecv addressed to THIS rank,
 * ncclGroupEnd - through the very code every halo exchange runs (run_ops on the side stream, evCompute -> side stream -> evComm ->
 * `hip_stream`, or with `deferred` != 0 the event a deferred group is awaited through): `bytes` bytes of device memory `src` arrive in
 * `dst` behind everything enqueued on `hip_stream` so far, and `hip_stream` waits for them. Needs nrdhip_tiler_rccl_init (a world of 1
 * is fine: the communicator then has one rank). tests/test_rccl_loopback.py; the sample has no counterpart (config 5 of BASELINE.json) */
NRDHIP_API int nrdhip_tiler_rccl_loopback(nrdhip_tiler* tiler, const void* src, void* dst, size_t bytes, int deferred, void* hip_stream);
NRDHIP_API int nrdhip_tiler_halo(nrdhip_tiler* tiler, uint32_t* rows);
/* out = {bytes sent, dispatches run as strips + interior, exchanges waited for in-frame, deferred exchanges} */
NRDHIP_API int nrdhip_tiler_stats(nrdhip_tiler* tiler, uint64_t out[4]);
NRDHIP_API const char* nrdhip_tiler_last_error(nrdhip_tiler* tiler);

/* ===================================================================================================================
 * Sample-side passes either side of the denoiser (SURVEY.md 8f "next" rows). Stand-alone: no instance, caller-owned
 * planes, work enqueued on `hip_stream`.
 * =================================================================================================================== */

/* History-confidence producer = the sample's "History confidence - Blur" loop: 5 dispatches of
 * Shaders/ConfidenceBlur.cs.hlsl:33-106 with step = 1..5, ping -> pong -> ping ... (Source/NRDSample.cpp:3999-4026);
 * the 5th lands in `pong`, the texture the sample binds to IN_DIFF_CONFIDENCE / IN_SPEC_CONFIDENCE (:457, :462).
 * Texel = RGBA16_SFLOAT {gradient, octahedral view normal xy, viewZ * FP16_VIEWZ_SCALE} (Shaders/SharcUpdate.cs.hlsl:249). */
typedef struct nrdhip_confidence_blur_desc {
    void* ping;               /* Gradient_Ping: input of the first pass (overwritten by the even passes) */
    void* pong;               /* Gradient_Pong: result */
    uint32_t pitch_bytes;     /* row pitch of both */
    uint16_t width, height;   /* Sample::GetSharcDims() (Source/NRDSample.cpp:596-598) */
    float camera_frustum[4];  /* gCameraFrustum (:3708) */
    float inv_size[2];        /* gInvSharcRenderSize (:3724) */
    float rect_width;         /* gRectSize.x */
    float unproject;          /* gUnproject (:3738) */
    float ortho_mode;         /* gOrthoMode */
    uint32_t frame_index;     /* gFrameIndex */
    uint32_t max_accumulated_frame_num; /* gMaxAccumulatedFrameNum (:3747) */
    uint32_t relax;           /* gDenoiserType == DENOISER_RELAX */
    uint32_t first_pass;      /* 0-based; 0 */
    uint32_t passes_num;      /* 5 (the sample's loop); sub-ranges allow per-pass timing */
} nrdhip_confidence_blur_desc;
NRDHIP_API int nrdhip_confidence_blur(const nrdhip_confidence_blur_desc* desc, void* hip_stream);

/* Back-end consumer = the NRD-facing part of Shaders/Composition.cs.hlsl:57-64 (shadow), :74-175 (diffuse / specular):
 * decodes OUT_* planes into linear radiance {rgb, normalised hit distance} and shadow {x, yzw}. */
enum { NRDHIP_UNPACK_NORMAL = 0, NRDHIP_UNPACK_OCCLUSION = 1, NRDHIP_UNPACK_SH = 2 };
typedef struct nrdhip_unpack_desc {
    uint16_t width, height;
    uint32_t mode;            /* NRDHIP_UNPACK_* (NRD_MODE of Shaders/Shared.hlsli) */
    uint32_t relax;           /* RELAX_BackEnd_* instead of REBLUR_BackEnd_* (Composition.cs.hlsl:160-166, 177-181) */
    uint32_t resolve;         /* SH mode: 1 = resolve against the G-buffer normal (gResolve), 0 = NRD_SG_ExtractColor */
    const void* diff; uint32_t diff_pitch;         /* OUT_DIFF_RADIANCE_HITDIST | OUT_DIFF_SH0 (RGBA16F) | OUT_DIFF_HITDIST (R16_UNORM); NULL = none */
    const void* spec; uint32_t spec_pitch;
    const void* diff_sh1; uint32_t diff_sh1_pitch; /* SH mode: OUT_DIFF_SH1 / OUT_SPEC_SH1 */
    const void* spec_sh1; uint32_t spec_sh1_pitch;
    const void* normal_roughness; uint32_t normal_roughness_pitch; /* IN_NORMAL_ROUGHNESS (R10G10B10A2), SH resolve only */
    const void* shadow; uint32_t shadow_pitch; uint32_t shadow_bytes_per_texel; /* OUT_SHADOW_TRANSLUCENCY RGBA8 (4) or R8 (1) */
    void* out_diff; uint32_t out_diff_pitch;       /* RGBA16F {linear rgb, hit distance term}; NULL = skip */
    void* out_spec; uint32_t out_spec_pitch;
    void* out_shadow; uint32_t out_shadow_pitch;   /* RGBA16F SIGMA_BackEnd_UnpackShadow result */
    float view_to_world[9];   /* SH resolve: rotation part of gViewToWorld (row-major 3x3) */
    float camera_frustum[4];
    float inv_rect_size[2];
} nrdhip_unpack_desc;
NRDHIP_API int nrdhip_backend_unpack(const nrdhip_unpack_desc* desc, void* hip_stream);

/* Producer side = what Shaders/TraceOpaque.cs.hlsl does with a path-tracing result before it becomes denoiser input (:421 hit
 * distance normalisation, :657 normal / roughness / materialID pack, :738-757 radiance texels per NRD_MODE for REBLUR / RELAX,
 * :800-801 SIGMA inputs), built on the host+device helper API include/nrd_frontend.h. Raw fp32 planes in (RGBA32F unless noted),
 * NRD input planes out; NULL skips a plane. */
enum { NRDHIP_PACK_DIRECTIONAL_OCCLUSION = 3 }; /* `mode` = NRDHIP_UNPACK_NORMAL / _OCCLUSION / _SH or this */
typedef struct nrdhip_frontend_pack_desc {
    uint16_t width, height;
    uint32_t mode;
    uint32_t relax;                      /* RELAX_FrontEnd_* (linear RGB, world-space hit distance) instead of REBLUR_FrontEnd_* */
    uint32_t sanitize;                   /* USE_SANITIZATION: non-finite radiance / hit distance -> 0 */
    float hit_distance_parameters[4];    /* nrd::ReblurHitDistanceParameters {A, B, C, D} (gHitDistSettings) */
    float tan_of_light_angular_radius;   /* gTanSunAngularRadius */
    const void* normal; uint32_t normal_pitch;                 /* {world normal xyz, linear roughness} */
    const void* material_id; uint32_t material_id_pitch;       /* R32F, 0..3 (optional) */
    const void* viewz; uint32_t viewz_pitch;                   /* R32F linear view depth (REBLUR hit distance normalisation) */
    const void* diff; uint32_t diff_pitch;                     /* {diffuse radiance rgb (demodulated), hit distance in world units} */
    const void* spec; uint32_t spec_pitch;
    const void* diff_direction; uint32_t diff_direction_pitch; /* SH / DIRECTIONAL_OCCLUSION: unit vector toward the light sample */
    const void* spec_direction; uint32_t spec_direction_pitch;
    const void* shadow; uint32_t shadow_pitch;                 /* {distance to occluder (>= 65504: miss), translucency rgb} */
    void* out_normal_roughness; uint32_t out_normal_roughness_pitch; /* IN_NORMAL_ROUGHNESS R10_G10_B10_A2_UNORM */
    void* out_diff; uint32_t out_diff_pitch;                   /* IN_DIFF_RADIANCE_HITDIST | _SH0 | _DIRECTION_HITDIST RGBA16F; OCCLUSION: IN_DIFF_HITDIST R16_UNORM */
    void* out_spec; uint32_t out_spec_pitch;
    void* out_diff_sh1; uint32_t out_diff_sh1_pitch;           /* SH mode: IN_DIFF_SH1 / IN_SPEC_SH1 RGBA16F */
    void* out_spec_sh1; uint32_t out_spec_sh1_pitch;
    void* out_penumbra; uint32_t out_penumbra_pitch;           /* IN_PENUMBRA R16F */
    void* out_translucency; uint32_t out_translucency_pitch;   /* IN_TRANSLUCENCY RGBA8 */
} nrdhip_frontend_pack_desc;
NRDHIP_API int nrdhip_frontend_pack(const nrdhip_frontend_pack_desc* desc, void* hip_stream);

/* Consumer side after the NRD decode = Shaders/Composition.cs.hlsl:92-107 (SH mode: NRD_SG_ReJitter against the 4 neighbours'
 * normals / depths) and :183-188 (material re-modulation: NRD_MaterialFactors from base colour / metalness; hair keeps factor 1,
 * Shaders/RaytracingShared.hlsli:925-936). Input = the planes nrdhip_backend_unpack wrote. */
typedef struct nrdhip_compose_desc {
    uint16_t width, height;
    uint32_t sh;                         /* 1: apply the re-jitter scales computed from the OUT_*_SH0 / SH1 planes below */
    uint32_t relax;
    uint32_t hair_material_id;           /* materialID whose factors stay 1 (MATERIAL_ID_HAIR); 0xffffffff = none */
    const void* diff; uint32_t diff_pitch;                     /* RGBA16F {linear radiance, hit distance term} from nrdhip_backend_unpack */
    const void* spec; uint32_t spec_pitch;
    const void* diff_sh0; uint32_t diff_sh0_pitch;             /* SH mode: the denoiser's OUT_DIFF_SH0 / SH1, OUT_SPEC_SH0 / SH1 */
    const void* diff_sh1; uint32_t diff_sh1_pitch;
    const void* spec_sh0; uint32_t spec_sh0_pitch;
    const void* spec_sh1; uint32_t spec_sh1_pitch;
    const void* normal_roughness; uint32_t normal_roughness_pitch;         /* IN_NORMAL_ROUGHNESS */
    const void* viewz; uint32_t viewz_pitch;                               /* IN_VIEWZ (re-jitter depth test) */
    const void* base_color_metalness; uint32_t base_color_metalness_pitch; /* RGBA8: sRGB base colour, metalness (gIn_BaseColor_Metalness); NULL = factors 1 */
    void* out_diff; uint32_t out_diff_pitch;                   /* RGBA16F Ldiff = diff x diffFactor (w passes through) */
    void* out_spec; uint32_t out_spec_pitch;
    float view_to_world[9];
    float camera_frustum[4];
    float inv_rect_size[2];
} nrdhip_compose_desc;
NRDHIP_API int nrdhip_compose(const nrdhip_compose_desc* desc, void* hip_stream);

/* Temporal anti-aliasing = Shaders/Taa.cs.hlsl:11-159 (one 16x16-group dispatch over the rect): 20x20 LDS tiles of the tonemapped
 * colour and of the motion vectors, 3x3 / 5x5 Gaussian moments, closest-depth motion vector, bicubic (5-tap "no corners")
 * history fetch, AABB clip + CIELAB-distance driven mix rate. */
typedef struct nrdhip_taa_desc {
    const void* mv; uint32_t mv_pitch;             /* gIn_Mv RGBA16F: xy = motion in pixels, w = viewZ * FP16_VIEWZ_SCALE * (+-1), < 0 asks for 5x5 */
    const void* composed; uint32_t composed_pitch; /* gIn_Composed RGBA16F (rgb) */
    const void* history; uint32_t history_pitch;   /* gIn_History RGBA16F {rgb, mix rate} = last frame's result */
    void* result; uint32_t result_pitch;           /* gOut_Result RGBA16F {rgb, mix rate} */
    uint16_t rect_width, rect_height;              /* gRectSize */
    uint16_t rect_width_prev, rect_height_prev;    /* gRectSizePrev */
    uint16_t render_width, render_height;          /* 1 / gInvRenderSize: size of the history texture (Source/NRDSample.cpp:3721) */
    uint32_t tonemap;                              /* ApplyTonemap active (NRD_MODE < OCCLUSION and gOnScreen <= SHOW_DENOISED_SPECULAR, Shared.hlsli:337-347) */
    float hdr_scale;                               /* gHdrScale (:3743) */
    float taa;                                     /* gTAA: lower bound of the mix rate (:3742) */
} nrdhip_taa_desc;
NRDHIP_API int nrdhip_taa(const nrdhip_taa_desc* desc, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* NRDHIP_H */
