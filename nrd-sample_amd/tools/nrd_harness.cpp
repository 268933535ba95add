// nrd_harness.cpp - headless C++ twin of the NRD part of the reference sample, written against include/NRD*.h exactly the
// way Source/NRDSample.cpp is written against the reference's headers:
//   Sample::Initialize .... denoiser descs + Integration::Recreate            (Source/NRDSample.cpp:869-990)
//   Sample::RenderFrame ... CommonSettings fill, NewFrame, SetCommonSettings  (:3835-3879)
//                           shadow denoising (:4068-4084), opaque denoising (:4086-4126), reference accumulation (:4213-4227)
//   Sample::Denoise ....... ResourceSnapshot::SetResource for every slot, Integration::Denoise (:440-531)
// Inputs are raw plane files (written by tests/test_cpp_harness.py or any producer); outputs are written back as raw files.
//
//   nrd_harness <dir> <width> <height> <frames> [--ranks N [--rccl | --async [--latency-us L] [--solo R]]] [--confidence] [--sh]
// --async (with --ranks): the in-process fabric as a stream-ordered transport - the tiler then runs its RCCL ordering path (side stream and
//   events) with device-to-device copies standing in for ncclSend / ncclRecv: what can be verified of that path on a 1-GPU box.
// --latency-us L (with --async): every exchange group of the stream-ordered transport starts with a kernel that spins L microseconds on the
//   stream the tiler hands it - a stand-in for the xGMI / RCCL latency this 1-GPU box does not have. Results must not change (ordering).
// --solo R (with --ranks N --async): ONLY rank R runs, against a loopback transport (its sends land in a scratch buffer, its receives copy
//   that buffer back after the injected latency): TIMING ONLY - the halo rows hold this rank's own data, outputs are not written. With the
//   GPU to itself the rank's frame time shows whether the strips-first schedule really hides an exchange behind the interior of its
//   dispatch (evCompute / evComm / evDeferred ordering): tests/test_cpp_harness.py compares frame times at L = 0 and L > 0.
// --confidence: the history-confidence path of the sample (Source/NRDSample.cpp:3999-4026, :457, :462, :3866): gradient.bin (RGBA16F at
//   Sample::GetSharcDims(), :596-598) goes through the five ConfidenceBlur passes - Gradient_Ping -> Pong -> Ping ..., the loop of
//   :4003-4026 - every frame, Gradient_Pong is bound to IN_DIFF_CONFIDENCE and IN_SPEC_CONFIDENCE, isHistoryConfidenceAvailable = true.
// --sh: NRD_MODE == SH (:464-476): REBLUR_DIFFUSE_SPECULAR_SH with the eight SH slots bound (diff_sh1.bin / spec_sh1.bin are the SH1
//   inputs, out_diff_sh1.bin / out_spec_sh1.bin are written).
// --ranks N: the same frames row-tiled across N ranks (nrd::TiledIntegration over the C++ row tiler, nrdhip_tiler_*), one host
// thread per rank; without --rccl the ranks share GPU 0 and halo rows travel through an in-process mailbox (a stand-in fabric:
// exercises the exchange plan, the strips / interior split and the band addressing on a 1-GPU box); with --rccl rank r runs on
// GPU r and the rows travel with ncclSend / ncclRecv (needs N GPUs: refuses loudly otherwise). Outputs are written exactly like
// the 1-rank run's, so the two can be compared byte for byte.
#include "../../include/NRDIntegration.h"

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define NRD_ID(x) nrd::Identifier(nrd::Denoiser::x) // same macro as the sample (Source/NRDSample.cpp:226)

struct Texture {
    void* ptr = nullptr;
    uint32_t pitch = 0;
    nrd::Format format = nrd::Format::MAX_NUM;
    uint16_t w = 0, h = 0;
    size_t bytes() const { return (size_t)pitch * h; }
};

static bool loadPlane(const std::string& path, Texture& t, bool optional = false) {
    std::vector<uint8_t> host(t.bytes(), 0);
    FILE* f = fopen(path.c_str(), "rb");
    if (f) {
        size_t n = fread(host.data(), 1, host.size(), f);
        fclose(f);
        if (n != host.size()) {
            fprintf(stderr, "short read: %s\n", path.c_str());
            return false;
        }
    } else if (!optional) {
        fprintf(stderr, "missing input: %s\n", path.c_str());
        return false;
    }
    return hipMemcpy(t.ptr, host.data(), host.size(), hipMemcpyHostToDevice) == hipSuccess;
}

static bool savePlane(const std::string& path, const Texture& t) {
    std::vector<uint8_t> host(t.bytes());
    if (hipMemcpy(host.data(), t.ptr, host.size(), hipMemcpyDeviceToHost) != hipSuccess)
        return false;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f)
        return false;
    fwrite(host.data(), 1, host.size(), f);
    fclose(f);
    return true;
}

static Texture makeTexture(uint16_t w, uint16_t h, nrd::Format fmt, uint32_t bpt) {
    Texture t;
    t.w = w;
    t.h = h;
    t.format = fmt;
    t.pitch = w * bpt;
    if (hipMalloc(&t.ptr, t.bytes()) != hipSuccess)
        t.ptr = nullptr;
    else
        (void)hipMemset(t.ptr, 0, t.bytes());
    return t;
}

static nrd::Resource GetNrdResource(const Texture& t) { // Sample::GetNrdResource (:416-438)
    nrd::Resource r = {};
    r.hip.ptr = t.ptr;
    r.hip.pitchBytes = t.pitch;
    r.hip.format = t.format;
    r.hip.width = t.w;
    r.hip.height = t.h;
    return r;
}

// ---- row-tiled run (--ranks N) -----------------------------------------------------------------------------------------
namespace tiled {

// in-process stand-in for the fabric: one FIFO per (source, destination) pair; sends never block, receives wait for their message
struct Mailbox {
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::vector<uint8_t>>> q;
};
struct Endpoint {
    Mailbox* box;
    int rank;
};
static int mbSend(void* user, const void* ptr, size_t bytes, int peer, void* stream) {
    auto* e = (Endpoint*)user;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) // the rows were produced on this stream
        return 1;
    std::vector<uint8_t> msg(bytes);
    if (hipMemcpy(msg.data(), ptr, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    {
        std::lock_guard<std::mutex> l(e->box->m);
        e->box->q[{e->rank, peer}].push_back(std::move(msg));
    }
    e->box->cv.notify_all();
    return 0;
}
static int mbRecv(void* user, void* ptr, size_t bytes, int peer, void*) {
    auto* e = (Endpoint*)user;
    std::vector<uint8_t> msg;
    {
        std::unique_lock<std::mutex> l(e->box->m);
        auto& dq = e->box->q[{peer, e->rank}];
        e->box->cv.wait(l, [&] { return !dq.empty(); });
        msg = std::move(dq.front());
        dq.pop_front();
    }
    if (msg.size() != bytes)
        return 1;
    return hipMemcpy(ptr, msg.data(), bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}

// --async: the same in-process fabric as a STREAM-ORDERED transport (NRDHIP_TRANSPORT_STREAM_ORDERED) - send / recv only enqueue work on
// the stream they are handed and return, like ncclSend / ncclRecv, so that the tiler takes its RCCL ordering path (side stream,
// evCompute -> side stream, evComm -> compute stream, evDeferred) on the one GPU of the box. A message is a staging buffer of a ring per
// (source, destination) pair: the sender copies its rows in on ITS stream and records `ready`; the receiver makes ITS stream wait for
// `ready`, copies the rows out and records `consumed`, which the sender's stream waits for before it overwrites the slot the next time
// round. The only host-side waits are for a message to have been POSTED (not for the GPU) and for a ring slot's reuse to be known.
// injected link latency: spins on the device for `ticks` of the constant-rate wall clock (hipDeviceAttributeWallClockRate, kHz)
__global__ void k_delay(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
}
static long long g_delayTicks = 0; // 0 = no injected latency
static void setLatencyUs(int us) {
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    g_delayTicks = (long long)us * (long long)khz / 1000;
}

// --solo: loopback transport of a single rank (timing only). A send lands in a scratch buffer per (peer, message index of the group), a
// receive copies the buffer of the same index back, behind the injected latency of its group - all on the stream the tiler hands over.
struct LoopbackEndpoint {
    std::map<std::pair<int, int>, std::pair<void*, size_t>> scratch; // (peer, index in group) -> buffer
    int sendIndex = 0, recvIndex = 0;
    bool delayed = false;
    uint64_t groups = 0;
};
static int lbBegin(void* user) {
    auto* e = (LoopbackEndpoint*)user;
    e->sendIndex = e->recvIndex = 0;
    e->delayed = false;
    e->groups++;
    return 0;
}
static void* lbBuffer(LoopbackEndpoint* e, int peer, int index, size_t bytes) {
    auto& b = e->scratch[{peer, index}];
    if (b.second < bytes) {
        if (b.first)
            (void)hipFree(b.first);
        if (hipMalloc(&b.first, bytes) != hipSuccess)
            return nullptr;
        (void)hipMemset(b.first, 0, bytes);
        b.second = bytes;
    }
    return b.first;
}
static int lbSend(void* user, const void* ptr, size_t bytes, int peer, void* stream) {
    auto* e = (LoopbackEndpoint*)user;
    void* b = lbBuffer(e, peer, e->sendIndex++, bytes);
    return b && hipMemcpyAsync(b, ptr, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : 1;
}
static int lbRecv(void* user, void* ptr, size_t bytes, int peer, void* stream) {
    auto* e = (LoopbackEndpoint*)user;
    hipStream_t st = (hipStream_t)stream;
    if (!e->delayed && g_delayTicks > 0) {
        hipLaunchKernelGGL(k_delay, dim3(1), dim3(1), 0, st, g_delayTicks);
        e->delayed = true;
    }
    void* b = lbBuffer(e, peer, e->recvIndex++, bytes);
    return b && hipMemcpyAsync(ptr, b, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess ? 0 : 1;
}

struct AsyncSlot {
    void* buf = nullptr;
    size_t cap = 0, bytes = 0;
    hipEvent_t ready = nullptr, consumed = nullptr;
    uint64_t posted = 0, taken = 0; // how many times the slot was filled / emptied (host bookkeeping under the fabric's mutex)
};
struct AsyncFabric {
    static constexpr int RING = 32;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::vector<AsyncSlot>> ring; // (source, destination) -> slots
    std::map<std::pair<int, int>, uint64_t> sent, received;     // messages posted / taken per pair
};
struct AsyncEndpoint {
    AsyncFabric* fab;
    int rank;
    bool delayed = false; // --latency-us: the first receive of a group carries the injected latency
};
static int asBegin(void* user) {
    ((AsyncEndpoint*)user)->delayed = false;
    return 0;
}
static int asSend(void* user, const void* ptr, size_t bytes, int peer, void* stream) {
    auto* e = (AsyncEndpoint*)user;
    hipStream_t st = (hipStream_t)stream;
    AsyncSlot* s;
    {
        std::unique_lock<std::mutex> l(e->fab->m);
        auto& v = e->fab->ring[{e->rank, peer}];
        if (v.empty())
            v.resize(AsyncFabric::RING);
        uint64_t& n = e->fab->sent[{e->rank, peer}];
        s = &v[n % AsyncFabric::RING];
        e->fab->cv.wait(l, [&] { return s->taken == s->posted; }); // the receiver has ENQUEUED its copy out of this slot's last use
        n++;
    }
    if (s->cap < bytes) { // (first rounds only; the plan repeats every frame)
        if (s->buf)
            (void)hipFree(s->buf);
        if (hipMalloc(&s->buf, bytes) != hipSuccess)
            return 1;
        s->cap = bytes;
    }
    if (!s->ready && (hipEventCreateWithFlags(&s->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->consumed, hipEventDisableTiming) != hipSuccess))
        return 1;
    if (s->posted && hipStreamWaitEvent(st, s->consumed, 0) != hipSuccess) // ... and that copy has run before the slot is overwritten
        return 1;
    if (hipMemcpyAsync(s->buf, ptr, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess || hipEventRecord(s->ready, st) != hipSuccess)
        return 1;
    {
        std::lock_guard<std::mutex> l(e->fab->m);
        s->bytes = bytes;
        s->posted++;
    }
    e->fab->cv.notify_all();
    return 0;
}
static int asRecv(void* user, void* ptr, size_t bytes, int peer, void* stream) {
    auto* e = (AsyncEndpoint*)user;
    hipStream_t st = (hipStream_t)stream;
    AsyncSlot* s;
    {
        std::unique_lock<std::mutex> l(e->fab->m);
        auto& v = e->fab->ring[{peer, e->rank}];
        if (v.empty())
            v.resize(AsyncFabric::RING);
        uint64_t& n = e->fab->received[{peer, e->rank}];
        s = &v[n % AsyncFabric::RING];
        const uint64_t want = n / AsyncFabric::RING + 1; // the slot's (n / RING + 1)-th fill is this message
        e->fab->cv.wait(l, [&] { return s->posted >= want; });
        n++;
        if (s->bytes != bytes)
            return 1;
    }
    if (!e->delayed && g_delayTicks > 0) {
        hipLaunchKernelGGL(k_delay, dim3(1), dim3(1), 0, st, g_delayTicks);
        e->delayed = true;
    }
    if (hipStreamWaitEvent(st, s->ready, 0) != hipSuccess || hipMemcpyAsync(ptr, s->buf, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipEventRecord(s->consumed, st) != hipSuccess)
        return 1;
    {
        std::lock_guard<std::mutex> l(e->fab->m);
        s->taken++;
    }
    e->fab->cv.notify_all();
    return 0;
}

struct Shared {
    std::string dir;
    uint16_t w, h;
    int frames, world;
    bool rccl;
    bool async = false;
    int solo = -1; // --solo R: only rank R runs, loopback transport, timing only
    AsyncFabric fabric;
    Mailbox box;
    uint8_t uniqueId[128];
    std::vector<uint8_t> outDiff, outSpec, outShadow, outSignal; // whole-frame outputs assembled from the ranks' owned rows
    std::vector<int> status;
};

static bool loadRows(const std::string& path, Texture& t, int row0, uint16_t frameH) {
    std::vector<uint8_t> host((size_t)t.pitch * frameH, 0);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "missing input: %s\n", path.c_str());
        return false;
    }
    size_t n = fread(host.data(), 1, host.size(), f);
    fclose(f);
    if (n != host.size())
        return false;
    return hipMemcpy(t.ptr, host.data() + (size_t)row0 * t.pitch, t.bytes(), hipMemcpyHostToDevice) == hipSuccess;
}
static bool fetchOwned(const Texture& t, int ownFirst, int ownRows, int globalRow, std::vector<uint8_t>& whole) {
    return hipMemcpy(whole.data() + (size_t)globalRow * t.pitch, (const uint8_t*)t.ptr + (size_t)ownFirst * t.pitch, (size_t)ownRows * t.pitch, hipMemcpyDeviceToHost) == hipSuccess;
}

static void rankMain(Shared* S, int rank) {
    using F = nrd::Format;
    using RT = nrd::ResourceType;
    int& status = S->status[rank];
    status = 1;
    const int device = S->rccl ? rank : 0;
    if (hipSetDevice(device) != hipSuccess)
        return;
    const uint16_t w = S->w, h = S->h;
    const nrd::DenoiserDesc denoisersDescs[] = {
        {NRD_ID(REBLUR_DIFFUSE_SPECULAR), nrd::Denoiser::REBLUR_DIFFUSE_SPECULAR},
        {NRD_ID(SIGMA_SHADOW), nrd::Denoiser::SIGMA_SHADOW_TRANSLUCENCY},
        {NRD_ID(REFERENCE), nrd::Denoiser::REFERENCE},
    };
    nrd::InstanceCreationDesc instanceCreationDesc = {};
    instanceCreationDesc.denoisers = denoisersDescs;
    instanceCreationDesc.denoisersNum = 3;
    nrd::IntegrationCreationDesc desc = {};
    desc.resourceWidth = w;
    desc.resourceHeight = h; // the WHOLE frame; the band follows from (rank, world, halo)
    // halo rows: reach of every pass with the settings used below + motion margin (static camera: the default 8 rows)
    uint32_t halo = 0;
    {
        nrd::Integration probe;
        nrd::IntegrationCreationDesc pd = desc;
        pd.resourceWidth = pd.resourceHeight = 64;
        if (probe.Recreate(pd, instanceCreationDesc, device) != nrd::Result::SUCCESS)
            return;
        nrd::CommonSettings cs = {};
        cs.resourceSize[0] = cs.resourceSize[1] = cs.rectSize[0] = cs.rectSize[1] = cs.resourceSizePrev[0] = cs.resourceSizePrev[1] = cs.rectSizePrev[0] = cs.rectSizePrev[1] = 64;
        cs.viewToClipMatrix[0] = cs.viewToClipMatrix[5] = cs.viewToClipMatrix[11] = cs.viewToClipMatrixPrev[0] = cs.viewToClipMatrixPrev[5] = cs.viewToClipMatrixPrev[11] = 1.0f;
        for (int i : {0, 5, 10, 15})
            cs.worldToViewMatrix[i] = cs.worldToViewMatrixPrev[i] = 1.0f;
        probe.SetCommonSettings(cs);
        const nrd::Identifier all[] = {NRD_ID(SIGMA_SHADOW), NRD_ID(REBLUR_DIFFUSE_SPECULAR), NRD_ID(REFERENCE)};
        if (nrdhip_required_halo((nrdhip_instance*)probe.GetInstance(), all, 3, 8, &halo) != 0)
            return;
    }
    Endpoint ep{&S->box, rank};
    AsyncEndpoint aep{&S->fabric, rank};
    LoopbackEndpoint lep;
    nrdhip_transport tr{&ep, nullptr, mbSend, mbRecv, nullptr, 0u};
    if (S->async)
        tr = nrdhip_transport{&aep, asBegin, asSend, asRecv, nullptr, NRDHIP_TRANSPORT_STREAM_ORDERED};
    if (S->solo >= 0)
        tr = nrdhip_transport{&lep, lbBegin, lbSend, lbRecv, nullptr, NRDHIP_TRANSPORT_STREAM_ORDERED};
    nrd::TiledIntegration m_NRD;
    if (m_NRD.Recreate(desc, instanceCreationDesc, device, rank, S->world, halo, S->rccl ? nullptr : &tr) != nrd::Result::SUCCESS) {
        fprintf(stderr, "rank %d: Recreate failed (bands shorter than the %u-row halo?)\n", rank, halo);
        return;
    }
    if (S->rccl && m_NRD.InitRccl(S->uniqueId) != nrd::Result::SUCCESS) {
        fprintf(stderr, "rank %d: RCCL init failed: %s\n", rank, m_NRD.GetTilerError());
        return;
    }
    const uint16_t lh = m_NRD.LocalHeight();
    const int row0 = m_NRD.Row0();
    Texture Mv = makeTexture(w, lh, F::RGBA16_SFLOAT, 8), Normal_Roughness = makeTexture(w, lh, F::R10_G10_B10_A2_UNORM, 4), ViewZ = makeTexture(w, lh, F::R32_SFLOAT, 4);
    Texture Unfiltered_Diff = makeTexture(w, lh, F::RGBA16_SFLOAT, 8), Unfiltered_Spec = makeTexture(w, lh, F::RGBA16_SFLOAT, 8);
    Texture Diff = makeTexture(w, lh, F::RGBA16_SFLOAT, 8), Spec = makeTexture(w, lh, F::RGBA16_SFLOAT, 8);
    Texture Unfiltered_Penumbra = makeTexture(w, lh, F::R16_SFLOAT, 2), Unfiltered_Translucency = makeTexture(w, lh, F::RGBA8_UNORM, 4), Shadow = makeTexture(w, lh, F::RGBA8_UNORM, 4);
    Texture Composed = makeTexture(w, lh, F::RGBA16_SFLOAT, 8), Validation = makeTexture(w, lh, F::RGBA8_UNORM, 4);
    const std::string& dir = S->dir;
    if (!loadRows(dir + "/mv.bin", Mv, row0, h) || !loadRows(dir + "/normal_roughness.bin", Normal_Roughness, row0, h) || !loadRows(dir + "/viewz.bin", ViewZ, row0, h) ||
        !loadRows(dir + "/diff.bin", Unfiltered_Diff, row0, h) || !loadRows(dir + "/spec.bin", Unfiltered_Spec, row0, h) || !loadRows(dir + "/penumbra.bin", Unfiltered_Penumbra, row0, h) ||
        !loadRows(dir + "/translucency.bin", Unfiltered_Translucency, row0, h) || !loadRows(dir + "/signal.bin", Composed, row0, h))
        return;
    hipStream_t stream;
    (void)hipStreamCreate(&stream);
    nrd::ReblurSettings m_ReblurSettings = {};
    nrd::SigmaSettings m_SigmaSettings = {};
    nrd::ReferenceSettings m_ReferenceSettings = {};
    auto snapshot = [&](nrd::ResourceSnapshot& rs) {
        rs.SetResource(RT::IN_MV, GetNrdResource(Mv));
        rs.SetResource(RT::IN_NORMAL_ROUGHNESS, GetNrdResource(Normal_Roughness));
        rs.SetResource(RT::IN_VIEWZ, GetNrdResource(ViewZ));
        rs.SetResource(RT::OUT_VALIDATION, GetNrdResource(Validation));
        rs.SetResource(RT::IN_DIFF_RADIANCE_HITDIST, GetNrdResource(Unfiltered_Diff));
        rs.SetResource(RT::OUT_DIFF_RADIANCE_HITDIST, GetNrdResource(Diff));
        rs.SetResource(RT::IN_SPEC_RADIANCE_HITDIST, GetNrdResource(Unfiltered_Spec));
        rs.SetResource(RT::OUT_SPEC_RADIANCE_HITDIST, GetNrdResource(Spec));
        rs.SetResource(RT::IN_PENUMBRA, GetNrdResource(Unfiltered_Penumbra));
        rs.SetResource(RT::IN_TRANSLUCENCY, GetNrdResource(Unfiltered_Translucency));
        rs.SetResource(RT::OUT_SHADOW_TRANSLUCENCY, GetNrdResource(Shadow));
        rs.SetResource(RT::IN_SIGNAL, GetNrdResource(Composed));
        rs.SetResource(RT::OUT_SIGNAL, GetNrdResource(Composed));
    };
    hipEvent_t evT0 = nullptr, evT1 = nullptr;
    (void)hipEventCreate(&evT0);
    (void)hipEventCreate(&evT1);
    const int timedFrom = S->frames > 8 ? 4 : S->frames; // (frame time is reported for runs of more than 8 frames: the first 4 are warm-up)
    for (int frameIndex = 0; frameIndex < S->frames; frameIndex++) {
        if (frameIndex == timedFrom)
            (void)hipEventRecord(evT0, stream);
        nrd::CommonSettings commonSettings = {};
        float aspect = (float)w / (float)h;
        float proj[16] = {1, 0, 0, 0, 0, aspect, 0, 0, 0, 0, 1, 1, 0, 0, -0.05f, 0};
        float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        memcpy(commonSettings.viewToClipMatrix, proj, sizeof(proj));
        memcpy(commonSettings.viewToClipMatrixPrev, proj, sizeof(proj));
        memcpy(commonSettings.worldToViewMatrix, ident, sizeof(ident));
        memcpy(commonSettings.worldToViewMatrixPrev, ident, sizeof(ident));
        commonSettings.motionVectorScale[0] = 1.0f / float(w);
        commonSettings.motionVectorScale[1] = 1.0f / float(h);
        commonSettings.motionVectorScale[2] = 1.0f;
        commonSettings.resourceSize[0] = commonSettings.resourceSizePrev[0] = commonSettings.rectSize[0] = commonSettings.rectSizePrev[0] = w;
        commonSettings.resourceSize[1] = commonSettings.resourceSizePrev[1] = commonSettings.rectSize[1] = commonSettings.rectSizePrev[1] = h; // the whole frame
        commonSettings.viewZScale = 1.0f;
        commonSettings.denoisingRange = 100.0f;
        commonSettings.disocclusionThreshold = 0.01f;
        commonSettings.disocclusionThresholdAlternate = 0.1f;
        commonSettings.frameIndex = (uint32_t)frameIndex;
        commonSettings.accumulationMode = frameIndex == 0 ? nrd::AccumulationMode::CLEAR_AND_RESTART : nrd::AccumulationMode::CONTINUE;
        m_NRD.NewFrame();
        m_NRD.SetCommonSettings(commonSettings);
        m_SigmaSettings.lightDirection[0] = 0.0f;
        m_SigmaSettings.lightDirection[1] = 0.0f;
        m_SigmaSettings.lightDirection[2] = -1.0f;
        nrd::ReblurHitDistanceParameters hitDistanceParameters = {};
        hitDistanceParameters.A = 3.0f;
        m_ReblurSettings.hitDistanceParameters = hitDistanceParameters;
        const nrd::Identifier order[] = {NRD_ID(SIGMA_SHADOW), NRD_ID(REBLUR_DIFFUSE_SPECULAR), NRD_ID(REFERENCE)};
        const void* settings[] = {&m_SigmaSettings, &m_ReblurSettings, &m_ReferenceSettings};
        for (int k = 0; k < 3; k++) {
            m_NRD.SetDenoiserSettings(order[k], settings[k]);
            nrd::ResourceSnapshot rs = {};
            snapshot(rs);
            if (frameIndex == 0 && k == 0) { // inputs are static here: one halo refresh (a renderer would do it per frame)
                const RT in[] = {RT::IN_MV, RT::IN_NORMAL_ROUGHNESS, RT::IN_VIEWZ, RT::IN_DIFF_RADIANCE_HITDIST, RT::IN_SPEC_RADIANCE_HITDIST, RT::IN_PENUMBRA, RT::IN_TRANSLUCENCY};
                if (m_NRD.ExchangeInputs(in, 7, stream, rs) != nrd::Result::SUCCESS)
                    return;
            }
            if (m_NRD.Denoise(&order[k], 1, stream, rs) != nrd::Result::SUCCESS) {
                fprintf(stderr, "rank %d: Denoise failed: %s / %s\n", rank, m_NRD.GetTilerError(), m_NRD.GetLastError());
                return;
            }
        }
    }
    if (m_NRD.Finish(stream) != nrd::Result::SUCCESS || hipEventRecord(evT1, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return;
    const int own0 = m_NRD.OwnFirst(), ownN = m_NRD.OwnRows(), g0 = row0 + own0;
    if (S->frames > timedFrom) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, evT0, evT1);
        uint64_t ts[4] = {};
        nrdhip_tiler_stats(m_NRD.GetTiler(), ts);
        printf("rank %d: %.4f ms per frame over %d frames (GPU time on the compute stream, exchanges included), %.1f in-frame + %.1f deferred exchanges per frame, "
               "injected latency %lld ticks per exchange%s\n", rank, ms / (float)(S->frames - timedFrom), S->frames - timedFrom, (double)ts[2] / S->frames, (double)ts[3] / S->frames,
               g_delayTicks, S->solo >= 0 ? " [solo: loopback transport, timing only]" : "");
    }
    if (S->solo >= 0) {
        m_NRD.Destroy();
        status = 0;
        return;
    }
    if (!fetchOwned(Diff, own0, ownN, g0, S->outDiff) || !fetchOwned(Spec, own0, ownN, g0, S->outSpec) || !fetchOwned(Shadow, own0, ownN, g0, S->outShadow) ||
        !fetchOwned(Composed, own0, ownN, g0, S->outSignal))
        return;
    uint64_t st[4] = {};
    nrdhip_tiler_stats(m_NRD.GetTiler(), st);
    printf("rank %d: rows [%d, %d), halo %u, %llu bytes sent, %llu dispatches split into strips + interior\n", rank, g0, g0 + ownN, halo, (unsigned long long)st[0],
           (unsigned long long)st[1]);
    m_NRD.Destroy();
    status = 0;
}

static int run(const std::string& dir, uint16_t w, uint16_t h, int frames, int world, bool rccl, bool async, int solo, int latencyUs) {
    Shared S;
    S.async = async;
    S.solo = solo;
    S.dir = dir;
    S.w = w;
    S.h = h;
    S.frames = frames;
    S.world = world;
    S.rccl = rccl;
    S.status.assign(world, 1);
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) {
        fprintf(stderr, "Recreate failed: no HIP device (there is no CPU fallback)\n");
        return 1;
    }
    if (rccl) {
        if (devices < world) {
            fprintf(stderr, "SKIPPED: --rccl --ranks %d needs %d GPUs, %d visible (RCCL refuses two ranks on one device)\n", world, world, devices);
            return 77;
        }
        if (nrd::TiledIntegration::GetUniqueId(S.uniqueId) != nrd::Result::SUCCESS) {
            fprintf(stderr, "ncclGetUniqueId failed\n");
            return 1;
        }
    }
    if (latencyUs > 0)
        setLatencyUs(latencyUs);
    if (solo >= 0) { // one rank against the loopback transport: timing only, nothing to assemble
        if (solo >= world || !async) {
            fprintf(stderr, "--solo R needs --ranks N > R and --async\n");
            return 2;
        }
        std::thread t(rankMain, &S, solo);
        t.join();
        if (S.status[solo]) {
            fprintf(stderr, "rank %d failed\n", solo);
            return 1;
        }
        printf("row-tiled run: rank %d of %d alone, loopback stream-ordered transport, %ux%u, %d frames, %d us injected per exchange\n", solo, world, w, h, frames, latencyUs);
        return 0;
    }
    S.outDiff.assign((size_t)w * 8 * h, 0);
    S.outSpec.assign((size_t)w * 8 * h, 0);
    S.outShadow.assign((size_t)w * 4 * h, 0);
    S.outSignal.assign((size_t)w * 8 * h, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++)
        th.emplace_back(rankMain, &S, r);
    for (auto& t : th)
        t.join();
    for (int r = 0; r < world; r++)
        if (S.status[r]) {
            fprintf(stderr, "rank %d failed\n", r);
            return 1;
        }
    auto save = [&](const char* name, const std::vector<uint8_t>& v) {
        FILE* f = fopen((dir + "/" + name).c_str(), "wb");
        if (!f)
            return false;
        fwrite(v.data(), 1, v.size(), f);
        fclose(f);
        return true;
    };
    if (!save("out_diff.bin", S.outDiff) || !save("out_spec.bin", S.outSpec) || !save("out_shadow.bin", S.outShadow) || !save("out_signal.bin", S.outSignal))
        return 1;
    printf("row-tiled run: %d ranks, %s transport, %ux%u, %d frames\n", world,
           rccl ? "RCCL" : (async ? "in-process stream-ordered (side stream + events, the RCCL ordering path)" : "in-process mailbox"), w, h, frames);
    return 0;
}

} // namespace tiled

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: nrd_harness <dir> <width> <height> <frames> [--ranks N [--rccl | --async [--latency-us L] [--solo R]]] [--confidence] [--sh] [--synthesize]\n");
        return 2;
    }
    std::string dir = argv[1];
    uint16_t w = (uint16_t)atoi(argv[2]), h = (uint16_t)atoi(argv[3]);
    int frames = atoi(argv[4]);
    int ranks = 1;
    bool rccl = false, async = false;
    int solo = -1, latencyUs = 0;
    for (int i = 5; i < argc; i++) {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc)
            ranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--solo") && i + 1 < argc)
            solo = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--latency-us") && i + 1 < argc)
            latencyUs = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rccl"))
            rccl = true;
        else if (!strcmp(argv[i], "--async"))
            async = true;
    }
    bool synthesize = false, confidence = false, sh = false;
    for (int i = 5; i < argc; i++) {
        if (!strcmp(argv[i], "--synthesize"))
            synthesize = true;
        else if (!strcmp(argv[i], "--confidence"))
            confidence = true;
        else if (!strcmp(argv[i], "--sh"))
            sh = true;
    }
    if ((confidence || sh) && (ranks > 1 || synthesize)) {
        fprintf(stderr, "--confidence / --sh belong to the 1-rank run on input files\n");
        return 2;
    }
    if (ranks > 1)
        return tiled::run(dir, w, h, frames, ranks, rccl, async, solo, latencyUs);
    using F = nrd::Format;
    using RT = nrd::ResourceType;

    // ---- Sample::Initialize: REBLUR + SIGMA + REFERENCE in one instance ----
    const nrd::DenoiserDesc denoisersDescs[] = {
        {NRD_ID(REBLUR_DIFFUSE_SPECULAR), sh ? nrd::Denoiser::REBLUR_DIFFUSE_SPECULAR_SH : nrd::Denoiser::REBLUR_DIFFUSE_SPECULAR}, // NRD_MODE (:871-921)
        {NRD_ID(SIGMA_SHADOW), nrd::Denoiser::SIGMA_SHADOW_TRANSLUCENCY}, // SIGMA_VARIANT (:48-52)
        {NRD_ID(REFERENCE), nrd::Denoiser::REFERENCE},
    };
    nrd::InstanceCreationDesc instanceCreationDesc = {};
    instanceCreationDesc.denoisers = denoisersDescs;
    instanceCreationDesc.denoisersNum = 3;
    nrd::IntegrationCreationDesc desc = {};
    snprintf(desc.name, sizeof(desc.name), "NRD");
    desc.queuedFrameNum = 3;
    desc.enableWholeLifetimeDescriptorCaching = true;
    desc.resourceWidth = w;
    desc.resourceHeight = h;
    desc.autoWaitForIdle = false;
    nrd::Integration m_NRD;
    if (m_NRD.Recreate(desc, instanceCreationDesc, 0) != nrd::Result::SUCCESS) {
        fprintf(stderr, "Recreate failed\n");
        return 1;
    }
    printf("NRD: allocated %.2f Mb for REBLUR, SIGMA and REFERENCE denoisers\n", m_NRD.GetTotalMemoryUsageInMb());
    const nrd::LibraryDesc& lib = *nrd::GetLibraryDesc();
    printf("NRD v%u.%u.%u normalEncoding %u roughnessEncoding %u\n", lib.versionMajor, lib.versionMinor, lib.versionBuild, (unsigned)lib.normalEncoding, (unsigned)lib.roughnessEncoding);

    // ---- CreateResourcesAndDescriptors (:2914-3007) ----
    Texture Mv = makeTexture(w, h, F::RGBA16_SFLOAT, 8), Normal_Roughness = makeTexture(w, h, F::R10_G10_B10_A2_UNORM, 4), ViewZ = makeTexture(w, h, F::R32_SFLOAT, 4);
    Texture Unfiltered_Diff = makeTexture(w, h, F::RGBA16_SFLOAT, 8), Unfiltered_Spec = makeTexture(w, h, F::RGBA16_SFLOAT, 8);
    Texture Diff = makeTexture(w, h, F::RGBA16_SFLOAT, 8), Spec = makeTexture(w, h, F::RGBA16_SFLOAT, 8);
    Texture Unfiltered_Penumbra = makeTexture(w, h, F::R16_SFLOAT, 2), Unfiltered_Translucency = makeTexture(w, h, F::RGBA8_UNORM, 4), Shadow = makeTexture(w, h, F::RGBA8_UNORM, 4);
    Texture Composed = makeTexture(w, h, F::RGBA16_SFLOAT, 8), Validation = makeTexture(w, h, F::RGBA8_UNORM, 4);
    // SH mode: the second texel of every signal (:2998-3003)
    Texture Unfiltered_DiffSh, Unfiltered_SpecSh, DiffSh, SpecSh;
    if (sh) {
        Unfiltered_DiffSh = makeTexture(w, h, F::RGBA16_SFLOAT, 8), Unfiltered_SpecSh = makeTexture(w, h, F::RGBA16_SFLOAT, 8);
        DiffSh = makeTexture(w, h, F::RGBA16_SFLOAT, 8), SpecSh = makeTexture(w, h, F::RGBA16_SFLOAT, 8);
        if (!loadPlane(dir + "/diff_sh1.bin", Unfiltered_DiffSh) || !loadPlane(dir + "/spec_sh1.bin", Unfiltered_SpecSh))
            return 1;
    }
    // history confidence: Gradient_Ping / Gradient_Pong at Sample::GetSharcDims() = 16 * ((renderResolution / SHARC_DOWNSCALE + 15) / 16) (:596-598, :2990)
    const uint16_t sharcW = (uint16_t)(16u * ((w / 5u + 15u) / 16u)), sharcH = (uint16_t)(16u * ((h / 5u + 15u) / 16u));
    Texture Gradient_Ping, Gradient_Pong, Gradient_Input;
    if (confidence) {
        Gradient_Ping = makeTexture(sharcW, sharcH, F::RGBA16_SFLOAT, 8), Gradient_Pong = makeTexture(sharcW, sharcH, F::RGBA16_SFLOAT, 8);
        Gradient_Input = makeTexture(sharcW, sharcH, F::RGBA16_SFLOAT, 8); // what SharcUpdate would write into Gradient_Ping every frame
        if (!loadPlane(dir + "/gradient.bin", Gradient_Input))
            return 1;
    }
    if (synthesize) {
        // --synthesize: no input files. A small analytic "path tracer" runs on the host (tilted floor + back wall, hashed noise on
        // radiance and hit distances) and its raw fp32 results go through the producer kernel nrdhip_frontend_pack - the HIP twin of
        // what Shaders/TraceOpaque.cs.hlsl:421, :657, :738-757, :800-801 do - straight into the slots the denoisers read.
        const size_t n = (size_t)w * h;
        std::vector<float> normal(n * 4), mat(n), viewz(n), diff(n * 4), spec(n * 4), shadow(n * 4);
        auto hash = [](uint32_t x) {
            x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
            return (float)(x >> 8) * (1.0f / 16777216.0f);
        };
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w; x++) {
                size_t i = (size_t)y * w + x;
                bool floor = y > h / 2u;
                float z = floor ? 3.0f + 30.0f * (float)(h - y) / (float)h : 20.0f;
                viewz[i] = z;
                float nx = 0.0f, ny = floor ? 0.8f : 0.0f, nz = floor ? -0.6f : -1.0f;
                normal[4 * i + 0] = nx, normal[4 * i + 1] = ny, normal[4 * i + 2] = nz;
                normal[4 * i + 3] = floor ? 0.3f + 0.4f * (float)((x / 32u) & 1u) : 0.6f;
                mat[i] = floor ? 0.0f : 1.0f;
                float n0 = hash((uint32_t)i * 4u + 0u), n1 = hash((uint32_t)i * 4u + 1u), n2 = hash((uint32_t)i * 4u + 2u), n3 = hash((uint32_t)i * 4u + 3u);
                float d = 0.2f + 1.6f * n0 * n0, sp = 0.1f + 2.5f * n1 * n1 * n1;
                diff[4 * i + 0] = 0.45f * d, diff[4 * i + 1] = 0.55f * d, diff[4 * i + 2] = 0.75f * d, diff[4 * i + 3] = 0.5f + 4.0f * n2;
                spec[4 * i + 0] = 0.9f * sp, spec[4 * i + 1] = 0.8f * sp, spec[4 * i + 2] = 0.7f * sp, spec[4 * i + 3] = 1.0f + 30.0f * n3;
                bool lit = ((x / 48u + y / 48u) & 1u) != 0u;
                shadow[4 * i + 0] = lit ? 65504.0f : 2.0f + 6.0f * n2;
                shadow[4 * i + 1] = 0.9f, shadow[4 * i + 2] = 0.6f, shadow[4 * i + 3] = 0.3f;
            }
        auto upload = [&](const std::vector<float>& v) {
            void* p = nullptr;
            if (hipMalloc(&p, v.size() * 4) != hipSuccess || hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                return (void*)nullptr;
            return p;
        };
        nrdhip_frontend_pack_desc pd = {};
        pd.width = w;
        pd.height = h;
        pd.mode = NRDHIP_UNPACK_NORMAL;
        pd.sanitize = 1;
        const float hp[4] = {3.0f, 0.1f, 20.0f, -25.0f};
        memcpy(pd.hit_distance_parameters, hp, sizeof(hp));
        pd.tan_of_light_angular_radius = 0.00465f;
        pd.normal = upload(normal), pd.normal_pitch = w * 16u;
        pd.material_id = upload(mat), pd.material_id_pitch = w * 4u;
        pd.viewz = upload(viewz), pd.viewz_pitch = w * 4u;
        pd.diff = upload(diff), pd.diff_pitch = w * 16u;
        pd.spec = upload(spec), pd.spec_pitch = w * 16u;
        pd.shadow = upload(shadow), pd.shadow_pitch = w * 16u;
        pd.out_normal_roughness = Normal_Roughness.ptr, pd.out_normal_roughness_pitch = Normal_Roughness.pitch;
        pd.out_diff = Unfiltered_Diff.ptr, pd.out_diff_pitch = Unfiltered_Diff.pitch;
        pd.out_spec = Unfiltered_Spec.ptr, pd.out_spec_pitch = Unfiltered_Spec.pitch;
        pd.out_penumbra = Unfiltered_Penumbra.ptr, pd.out_penumbra_pitch = Unfiltered_Penumbra.pitch;
        pd.out_translucency = Unfiltered_Translucency.ptr, pd.out_translucency_pitch = Unfiltered_Translucency.pitch;
        if (!pd.normal || !pd.material_id || !pd.viewz || !pd.diff || !pd.spec || !pd.shadow || nrdhip_frontend_pack(&pd, nullptr) != 0 ||
            hipMemcpy(ViewZ.ptr, viewz.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(Composed.ptr, Unfiltered_Diff.ptr, Composed.bytes(), hipMemcpyDeviceToDevice) != hipSuccess) {
            fprintf(stderr, "synthesize / nrdhip_frontend_pack failed\n");
            return 1;
        }
        // the packed inputs are written out so that another driver can be fed the very same planes (tests/test_cpp_harness.py)
        if (!savePlane(dir + "/mv.bin", Mv) || !savePlane(dir + "/normal_roughness.bin", Normal_Roughness) || !savePlane(dir + "/viewz.bin", ViewZ) ||
            !savePlane(dir + "/diff.bin", Unfiltered_Diff) || !savePlane(dir + "/spec.bin", Unfiltered_Spec) || !savePlane(dir + "/penumbra.bin", Unfiltered_Penumbra) ||
            !savePlane(dir + "/translucency.bin", Unfiltered_Translucency) || !savePlane(dir + "/signal.bin", Composed))
            return 1;
        printf("synthesized inputs through nrdhip_frontend_pack\n");
    } else if (!loadPlane(dir + "/mv.bin", Mv) || !loadPlane(dir + "/normal_roughness.bin", Normal_Roughness) || !loadPlane(dir + "/viewz.bin", ViewZ) ||
               !loadPlane(dir + "/diff.bin", Unfiltered_Diff) || !loadPlane(dir + "/spec.bin", Unfiltered_Spec) || !loadPlane(dir + "/penumbra.bin", Unfiltered_Penumbra) ||
               !loadPlane(dir + "/translucency.bin", Unfiltered_Translucency) || !loadPlane(dir + "/signal.bin", Composed))
        return 1;

    nrd::ReblurSettings m_ReblurSettings = {};
    nrd::SigmaSettings m_SigmaSettings = {};
    nrd::ReferenceSettings m_ReferenceSettings = {};
    hipStream_t stream;
    (void)hipStreamCreate(&stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);

    auto Denoise = [&](const nrd::Identifier* denoisers, uint32_t denoiserNum) -> nrd::Result { // Sample::Denoise (:440-531)
        nrd::ResourceSnapshot resourceSnapshot = {};
        resourceSnapshot.restoreInitialState = false;
        resourceSnapshot.SetResource(RT::IN_MV, GetNrdResource(Mv));
        resourceSnapshot.SetResource(RT::IN_NORMAL_ROUGHNESS, GetNrdResource(Normal_Roughness));
        resourceSnapshot.SetResource(RT::IN_VIEWZ, GetNrdResource(ViewZ));
        resourceSnapshot.SetResource(RT::OUT_VALIDATION, GetNrdResource(Validation));
        resourceSnapshot.SetResource(RT::IN_DIFF_RADIANCE_HITDIST, GetNrdResource(Unfiltered_Diff));
        resourceSnapshot.SetResource(RT::OUT_DIFF_RADIANCE_HITDIST, GetNrdResource(Diff));
        resourceSnapshot.SetResource(RT::IN_SPEC_RADIANCE_HITDIST, GetNrdResource(Unfiltered_Spec));
        resourceSnapshot.SetResource(RT::OUT_SPEC_RADIANCE_HITDIST, GetNrdResource(Spec));
        if (confidence) { // one texture on both slots (:457, :462)
            resourceSnapshot.SetResource(RT::IN_DIFF_CONFIDENCE, GetNrdResource(Gradient_Pong));
            resourceSnapshot.SetResource(RT::IN_SPEC_CONFIDENCE, GetNrdResource(Gradient_Pong));
        }
        if (sh) { // NRD_MODE == SH (:464-476)
            resourceSnapshot.SetResource(RT::IN_DIFF_SH0, GetNrdResource(Unfiltered_Diff));
            resourceSnapshot.SetResource(RT::IN_DIFF_SH1, GetNrdResource(Unfiltered_DiffSh));
            resourceSnapshot.SetResource(RT::OUT_DIFF_SH0, GetNrdResource(Diff));
            resourceSnapshot.SetResource(RT::OUT_DIFF_SH1, GetNrdResource(DiffSh));
            resourceSnapshot.SetResource(RT::IN_SPEC_SH0, GetNrdResource(Unfiltered_Spec));
            resourceSnapshot.SetResource(RT::IN_SPEC_SH1, GetNrdResource(Unfiltered_SpecSh));
            resourceSnapshot.SetResource(RT::OUT_SPEC_SH0, GetNrdResource(Spec));
            resourceSnapshot.SetResource(RT::OUT_SPEC_SH1, GetNrdResource(SpecSh));
        }
        resourceSnapshot.SetResource(RT::IN_PENUMBRA, GetNrdResource(Unfiltered_Penumbra));
        resourceSnapshot.SetResource(RT::IN_TRANSLUCENCY, GetNrdResource(Unfiltered_Translucency));
        resourceSnapshot.SetResource(RT::OUT_SHADOW_TRANSLUCENCY, GetNrdResource(Shadow));
        resourceSnapshot.SetResource(RT::IN_SIGNAL, GetNrdResource(Composed));
        resourceSnapshot.SetResource(RT::OUT_SIGNAL, GetNrdResource(Composed)); // same texture, in place (:484-485)
        return m_NRD.Denoise(denoisers, denoiserNum, stream, resourceSnapshot);
    };

    float ms = 0.0f;
    for (int frameIndex = 0; frameIndex < frames; frameIndex++) {
        // ---- RenderFrame: NRD common settings (:3835-3876); static pinhole camera looking down +z
        nrd::CommonSettings commonSettings = {};
        float aspect = (float)w / (float)h;
        float proj[16] = {1, 0, 0, 0, 0, aspect, 0, 0, 0, 0, 1, 1, 0, 0, -0.05f, 0};
        float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        memcpy(commonSettings.viewToClipMatrix, proj, sizeof(proj));
        memcpy(commonSettings.viewToClipMatrixPrev, proj, sizeof(proj));
        memcpy(commonSettings.worldToViewMatrix, ident, sizeof(ident));
        memcpy(commonSettings.worldToViewMatrixPrev, ident, sizeof(ident));
        commonSettings.motionVectorScale[0] = 1.0f / float(w);
        commonSettings.motionVectorScale[1] = 1.0f / float(h);
        commonSettings.motionVectorScale[2] = 1.0f;
        commonSettings.resourceSize[0] = commonSettings.resourceSizePrev[0] = commonSettings.rectSize[0] = commonSettings.rectSizePrev[0] = w;
        commonSettings.resourceSize[1] = commonSettings.resourceSizePrev[1] = commonSettings.rectSize[1] = commonSettings.rectSizePrev[1] = h;
        commonSettings.viewZScale = 1.0f;
        commonSettings.denoisingRange = 100.0f;
        commonSettings.disocclusionThreshold = 0.01f;
        commonSettings.disocclusionThresholdAlternate = 0.1f;
        commonSettings.frameIndex = (uint32_t)frameIndex;
        commonSettings.accumulationMode = frameIndex == 0 ? nrd::AccumulationMode::CLEAR_AND_RESTART : nrd::AccumulationMode::CONTINUE;
        commonSettings.isHistoryConfidenceAvailable = confidence; // (:3866)
        if (confidence) {
            // "History confidence - Blur" (:3999-4026): five passes, step = 1 + i, ping -> pong -> ping ...; the fifth lands in Gradient_Pong
            if (hipMemcpyAsync(Gradient_Ping.ptr, Gradient_Input.ptr, Gradient_Ping.bytes(), hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return 1;
            nrdhip_confidence_blur_desc cb = {};
            cb.ping = Gradient_Ping.ptr, cb.pong = Gradient_Pong.ptr, cb.pitch_bytes = Gradient_Ping.pitch;
            cb.width = sharcW, cb.height = sharcH;
            const float frustum[4] = {-1.0f, -1.0f / aspect, 2.0f, 2.0f / aspect}; // gCameraFrustum of the projection above (x0, y0, dx, dy at z = 1)
            memcpy(cb.camera_frustum, frustum, sizeof(frustum));
            cb.inv_size[0] = 1.0f / (float)sharcW, cb.inv_size[1] = 1.0f / (float)sharcH;
            cb.rect_width = (float)w;
            cb.unproject = 1.0f / (0.5f * (float)h * aspect);
            cb.ortho_mode = 0.0f;
            cb.frame_index = (uint32_t)frameIndex;
            cb.max_accumulated_frame_num = 30;
            cb.relax = 0;
            for (uint32_t i = 0; i < 5u; i++) { // must be odd
                cb.first_pass = i, cb.passes_num = 1;
                if (nrdhip_confidence_blur(&cb, stream) != 0) {
                    fprintf(stderr, "nrdhip_confidence_blur failed\n");
                    return 1;
                }
            }
        }

        m_NRD.NewFrame();
        m_NRD.SetCommonSettings(commonSettings);
        if (frameIndex == frames - 1)
            (void)hipEventRecord(e0, stream);
        { // Shadow denoising (:4068-4084)
            m_SigmaSettings.lightDirection[0] = 0.0f;
            m_SigmaSettings.lightDirection[1] = 0.0f;
            m_SigmaSettings.lightDirection[2] = -1.0f;
            nrd::Identifier denoiser = NRD_ID(SIGMA_SHADOW);
            m_NRD.SetDenoiserSettings(denoiser, &m_SigmaSettings);
            if (Denoise(&denoiser, 1) != nrd::Result::SUCCESS)
                return 1;
        }
        { // Opaque denoising (:4086-4126)
            nrd::ReblurHitDistanceParameters hitDistanceParameters = {};
            hitDistanceParameters.A = 3.0f;
            m_ReblurSettings.hitDistanceParameters = hitDistanceParameters;
            nrd::ReblurSettings settings = m_ReblurSettings;
            const nrd::Identifier denoisers[] = {NRD_ID(REBLUR_DIFFUSE_SPECULAR)};
            m_NRD.SetDenoiserSettings(denoisers[0], &settings);
            if (Denoise(denoisers, 1) != nrd::Result::SUCCESS)
                return 1;
        }
        { // Reference accumulation (:4213-4227)
            nrd::Identifier denoiser = NRD_ID(REFERENCE);
            m_NRD.SetDenoiserSettings(denoiser, &m_ReferenceSettings);
            if (Denoise(&denoiser, 1) != nrd::Result::SUCCESS) {
                fprintf(stderr, "Denoise failed: %s\n", m_NRD.GetLastError());
                return 1;
            }
        }
        if (frameIndex == frames - 1)
            (void)hipEventRecord(e1, stream);
    }
    if (hipStreamSynchronize(stream) != hipSuccess) {
        fprintf(stderr, "stream failed: %s\n", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("last frame: %.3f ms for SIGMA + REBLUR_DIFFUSE_SPECULAR + REFERENCE at %ux%u\n", ms, w, h);
    if (!savePlane(dir + "/out_diff.bin", Diff) || !savePlane(dir + "/out_spec.bin", Spec) || !savePlane(dir + "/out_shadow.bin", Shadow) || !savePlane(dir + "/out_signal.bin", Composed))
        return 1;
    if (sh && (!savePlane(dir + "/out_diff_sh1.bin", DiffSh) || !savePlane(dir + "/out_spec_sh1.bin", SpecSh)))
        return 1;
    if (confidence && !savePlane(dir + "/confidence.bin", Gradient_Pong))
        return 1;
    m_NRD.Destroy(); // before the device goes away (:744-748)
    return 0;
}
