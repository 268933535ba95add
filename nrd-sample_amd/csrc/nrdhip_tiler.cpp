// nrdhip_tiler.cpp - row tiling of one frame across the GPUs of a node, below the C-ABI (include/nrdhip.h "row tiler").
//
// One process (or thread) per GPU. Rank r owns a contiguous band of rows of the global frame and keeps `halo` extra rows of every
// plane on each side (nrdhip_create_desc band fields). After every recorded dispatch the rows a row neighbour may read are sent
// to that neighbour - point to point, each pair on its own xGMI link, no collective and no ring (SURVEY.md 8e scheme A
// "per-pass halo"). The reference has no counterpart (single adapter, single queue: Source/NRDSample.cpp:755-778).
//
// This layer sits on top of the GetComputeDispatches-style part of the C-ABI only (nrdhip_dispatch_info_get, nrdhip_denoise_range,
// nrdhip_denoise_rows, nrdhip_pool_info): it derives the exchange plan from what each dispatch reads / writes and how far it
// reads (`halo_rows`), and REFUSES a dispatch list that reads farther than the rows the band stores - the tiled result is
// bit-identical to a single-GPU run or the call fails, never silently different.
//
// Transport: RCCL (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a side stream, ordered against the compute stream with
// events) - resolved from librccl at nrdhip_tiler_rccl_init so that the library loads on hosts without it - or caller-supplied
// callbacks (tests: host memcpy / gloo; a host with its own fabric layer).
//
// Overlap: when the band is tall enough a dispatch first runs on the boundary strips the neighbours need (nrdhip_denoise_rows),
// the exchange of those rows starts on the side stream behind an event, the interior runs meanwhile, and only the next dispatch
// waits for the rows. Rows that only the NEXT frame reads (permanent planes surviving the frame: history, accumulation speeds,
// the guide ...) follow without strips and without a wait and are awaited when the next dispatch list starts.
#include "../../include/nrdhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h> // declarations only: librccl is resolved with dlopen at nrdhip_tiler_rccl_init, never linked

namespace {

struct Xfer { // rows of one pool plane: `rows` full-resolution rows next to each band edge, the `skip` nearest the edge already delivered
    uint32_t code, rows, skip;
    uint32_t prefix = (uint32_t)NRDHIP_NO_PLANE; // tap-texel plane: the guide plane its texels start with (nrdhip_dispatch_info::written_prefix)
};
struct Staging { // the signal halves of the tap texels of some rows, packed: what travels instead of the rows themselves
    const void* key;
    bool send;
    int peer;
    void* ptr;
    size_t bytes;
};
struct PlanEntry {
    std::vector<Xfer> now, later;
};

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (lib)
            return true;
        // a process that already carries an RCCL (PyTorch bundles one) resolves to that copy: one RCCL per process
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib)
                break;
        }
        if (!lib) {
            err = std::string("librccl not loadable: ") + dlerror();
            return false;
        }
#define NRD_SYM(field, name)                                                   \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, name));               \
    if (!field) {                                                              \
        err = std::string("librccl lacks ") + name;                            \
        return false;                                                          \
    }
        NRD_SYM(GetUniqueId, "ncclGetUniqueId")
        NRD_SYM(CommInitRank, "ncclCommInitRank")
        NRD_SYM(CommDestroy, "ncclCommDestroy")
        NRD_SYM(GroupStart, "ncclGroupStart")
        NRD_SYM(GroupEnd, "ncclGroupEnd")
        NRD_SYM(Send, "ncclSend")
        NRD_SYM(Recv, "ncclRecv")
        NRD_SYM(GetErrorString, "ncclGetErrorString")
#undef NRD_SYM
        return true;
    }
};
RcclApi g_rccl;

} // namespace

namespace nrdhip {
bool instance_is_live(const nrdhip_instance* inst); // nrdhip.cpp
}

struct nrdhip_tiler {
    nrdhip_instance* inst = nullptr;
    int rank = 0, world = 1;
    uint32_t halo = 0;
    int32_t frameH = 0, row0 = 0, ownFirst = 0, ownRows = 0, localH = 0;
    nrdhip_transport tr{};
    bool custom = false;
    // custom transport whose callbacks complete the transfer before they return (host transports): no side stream, no events. RCCL and
    // stream-ordered custom transports (NRDHIP_TRANSPORT_STREAM_ORDERED) share ONE ordering path: side stream + the three events below
    bool hostOrdered() const { return custom && !(tr.flags & NRDHIP_TRANSPORT_STREAM_ORDERED); }
    ncclComm_t comm = nullptr;
    hipStream_t commStream = nullptr;
    hipEvent_t evCompute = nullptr, evComm = nullptr;
    // Rows only the NEXT frame reads (permanent planes that survive the frame) travel as ONE group behind the last dispatch of their list -
    // round 5: six groups per REBLUR frame, each queued IN FRONT of the next dispatch's strips-first exchange on the in-order side stream,
    // put 13 latencies on a frame's critical path where 7 belong (profiles/r05_exchange_overlap_solo.json) - and are awaited by the SAME
    // list's next call, right before its first dispatch that reads previous-frame state: another list's call (SIGMA, then REBLUR, then
    // REFERENCE inside one frame) does not wait for them unless it shares a denoiser with the sender (wait_deferred_sharing). One event per
    // identifier list, at most 16 lists (deferred_slot).
    struct DeferredSlot {
        std::vector<uint32_t> ids;
        hipEvent_t ev = nullptr;
        bool pending = false;
    };
    std::vector<DeferredSlot> deferred;
    uint32_t firstPrevRead = 0; // first dispatch of the planned list that reads a permanent plane before the list writes it
    std::vector<uint32_t> planIds;
    std::vector<PlanEntry> plan;
    std::vector<uint32_t> planSig; // planes + reach of the dispatches the plan was built from (build_plan)
    uint64_t bytesSent = 0, splitDispatches = 0, exchanges = 0, deferredExchanges = 0;
    uint32_t reprojRows = 0; // rows of previous-frame state exchanged per surviving permanent plane (build_plan)
    std::vector<Staging> staging;
    std::string error;
};

namespace {

[[maybe_unused]] const int FAILURE = 1, INVALID = 2, UNSUPPORTED = 3; // nrd::Result values (include/NRDDescs.h)

int fail(nrdhip_tiler& T, int code, const std::string& what) {
    T.error = what;
    return code;
}

// (pointer, rows, pitch) of pool plane `code` and the divisor between full-resolution rows and its rows (tile planes: 16)
bool plane_of(nrdhip_tiler& T, uint32_t code, nrdhip_plane_info& P, uint32_t& div) {
    if ((code >> 16) > 1 || nrdhip_pool_info(T.inst, code >> 16, code & 0xffff, &P) != 0 || !P.ptr || !P.height)
        return false;
    div = (int32_t)P.height == T.localH ? 1u : 16u; // pool planes are full resolution or one texel per 16x16 tile (rounding localH / height
    return true;                                    // went wrong for short bands: 449 rows over 29 tile rows is 15)
}

// the instance's HIP device made current for a call (streams, events and the RCCL communicator of the tiler must live on the GPU
// the pools live on), the caller's restored afterwards
struct TilerDeviceScope {
    int prev = -1;
    bool switched = false;
    explicit TilerDeviceScope(nrdhip_tiler& T) {
        const int device = nrdhip_get_device(T.inst);
        if (device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != device)
            switched = hipSetDevice(device) == hipSuccess;
    }
    ~TilerDeviceScope() {
        if (switched)
            (void)hipSetDevice(prev);
    }
};

struct Op {
    bool send;
    uint8_t* ptr;
    size_t bytes;
    int peer;
    // tap-texel rows (Xfer::prefix): ptr / bytes are the staging buffer; the signal halves are gathered from `plane` before a send and
    // scattered into it after a receive, which also puts the guide halves back from this rank's own guide plane
    uint8_t* plane = nullptr;
    const uint8_t* guide = nullptr;
    uint32_t pitch = 0, gpitch = 0, width = 0, rows = 0;
};

// one thread per texel of a run of rows: the last 8 bytes of every 16-byte tap texel -> staging (pack), staging + the same pixel's 8-byte
// guide texel -> the tap texel (unpack)
__global__ void k_tap_pack(const uint8_t* plane, uint32_t pitch, uint32_t width, uint2* staging) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, r = blockIdx.y;
    if (x < width)
        staging[(size_t)r * width + x] = *reinterpret_cast<const uint2*>(plane + (size_t)r * pitch + (size_t)x * 16u + 8u);
}
__global__ void k_tap_unpack(uint8_t* plane, uint32_t pitch, const uint8_t* guide, uint32_t gpitch, uint32_t width, const uint2* staging) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, r = blockIdx.y;
    if (x < width) {
        const uint2 g = *reinterpret_cast<const uint2*>(guide + (size_t)r * gpitch + (size_t)x * 8u), v = staging[(size_t)r * width + x];
        *reinterpret_cast<uint4*>(plane + (size_t)r * pitch + (size_t)x * 16u) = uint4{g.x, g.y, v.x, v.y};
    }
}

void* staging_of(nrdhip_tiler& T, const void* key, bool send, int peer, size_t bytes) {
    for (auto& st : T.staging)
        if (st.key == key && st.send == send && st.peer == peer && st.bytes == bytes)
            return st.ptr;
    void* ptr = nullptr;
    if (hipMalloc(&ptr, bytes) != hipSuccess)
        return nullptr;
    T.staging.push_back({key, send, peer, ptr, bytes});
    return ptr;
}

// send / recv descriptors of one plane: the `rows` owned rows next to each band edge go to that neighbour's halo; `skip` of them
// (nearest the edge) were delivered earlier. Order per peer is the same on both sides (NCCL matches sends and receives of a
// group by order): [send, recv] toward the upper neighbour, then [send, recv] toward the lower one.
bool ops_of(nrdhip_tiler& T, uint8_t* base, uint32_t pitch, uint32_t planeRows, uint32_t div, uint32_t rows, uint32_t skip, std::vector<Op>& ops,
            const uint8_t* guide = nullptr, uint32_t gpitch = 0, uint32_t width = 0) {
    const uint32_t hskip = skip / div, hrows = std::max<uint32_t>((rows + div - 1) / div, 1u);
    const uint32_t first = (uint32_t)T.ownFirst / div, n = std::max<uint32_t>((uint32_t)T.ownRows / div, 1u);
    bool ok = true;
    auto push = [&](bool send, uint32_t a, uint32_t b, int peer) {
        if (b <= a)
            return;
        if (!guide) {
            ops.push_back({send, base + (size_t)a * pitch, (size_t)(b - a) * pitch, peer});
            return;
        }
        Op o{send, nullptr, (size_t)(b - a) * width * 8u, peer};
        o.plane = base + (size_t)a * pitch;
        o.guide = guide + (size_t)a * gpitch;
        o.pitch = pitch;
        o.gpitch = gpitch;
        o.width = width;
        o.rows = b - a;
        o.ptr = (uint8_t*)staging_of(T, o.plane, send, peer, o.bytes);
        ok = ok && o.ptr;
        ops.push_back(o);
    };
    if (T.rank > 0) {
        push(true, first + hskip, first + std::min(hrows, n), T.rank - 1);
        push(false, first - std::min(hrows, first), first - std::min(hskip, first), T.rank - 1);
    }
    if (T.rank < T.world - 1) {
        push(true, first + n - std::min(hrows, n), first + n - std::min(hskip, n), T.rank + 1);
        push(false, std::min(first + n + hskip, planeRows), std::min(first + n + hrows, planeRows), T.rank + 1);
    }
    return ok;
}

// run one batch of transfers on `stream` (RCCL: one group; custom transport: its callbacks in the same order)
int run_ops(nrdhip_tiler& T, const std::vector<Op>& ops, hipStream_t stream) {
    if (ops.empty())
        return 0;
    for (auto& o : ops)
        if (o.send) {
            T.bytesSent += o.bytes;
            if (o.plane) // tap texels: gather the signal halves (same stream as the transfer, ahead of it)
                hipLaunchKernelGGL(k_tap_pack, dim3((o.width + 255u) / 256u, o.rows, 1), dim3(256, 1, 1), 0, stream, (const uint8_t*)o.plane, o.pitch, o.width, (uint2*)o.ptr);
        }
    auto unpack = [&]() { // ... and behind it: received halves + this rank's guide texels -> the halo rows
        for (auto& o : ops)
            if (!o.send && o.plane)
                hipLaunchKernelGGL(k_tap_unpack, dim3((o.width + 255u) / 256u, o.rows, 1), dim3(256, 1, 1), 0, stream, o.plane, o.pitch, o.guide, o.gpitch, o.width, (const uint2*)o.ptr);
        return 0;
    };
    if (T.custom) {
        if (T.tr.group_begin && T.tr.group_begin(T.tr.user) != 0)
            return fail(T, FAILURE, "transport group_begin failed");
        for (auto& o : ops) {
            int r = o.send ? T.tr.send(T.tr.user, o.ptr, o.bytes, o.peer, stream) : T.tr.recv(T.tr.user, o.ptr, o.bytes, o.peer, stream);
            if (r != 0)
                return fail(T, FAILURE, "transport send / recv failed");
        }
        if (T.tr.group_end && T.tr.group_end(T.tr.user, stream) != 0)
            return fail(T, FAILURE, "transport group_end failed");
        return unpack();
    }
    if (!T.comm)
        return fail(T, INVALID, "no transport: call nrdhip_tiler_rccl_init or pass callbacks to nrdhip_tiler_create");
    ncclResult_t r = g_rccl.GroupStart();
    for (auto& o : ops) {
        if (r != ncclSuccess)
            break;
        r = o.send ? g_rccl.Send(o.ptr, o.bytes, ncclUint8, o.peer, T.comm, stream) : g_rccl.Recv(o.ptr, o.bytes, ncclUint8, o.peer, T.comm, stream);
    }
    ncclResult_t e = g_rccl.GroupEnd();
    if (r == ncclSuccess)
        r = e;
    if (r != ncclSuccess)
        return fail(T, FAILURE, std::string("RCCL: ") + g_rccl.GetErrorString(r));
    return unpack();
}

int collect(nrdhip_tiler& T, const std::vector<Xfer>& list, std::vector<Op>& ops) {
    for (auto& x : list) {
        nrdhip_plane_info P, G{};
        uint32_t div, gdiv = 1;
        if (!plane_of(T, x.code, P, div))
            return fail(T, INVALID, "pool plane of the exchange plan is not bound");
        const bool tap = x.prefix != (uint32_t)NRDHIP_NO_PLANE;
        if (tap && (!plane_of(T, x.prefix, G, gdiv) || div != 1 || gdiv != 1 || P.bytes_per_texel != 16 || G.bytes_per_texel != 8 || G.height != P.height))
            return fail(T, INVALID, "tap-texel plane and its guide plane do not match (16 / 8 bytes per texel, same rows)");
        if (!ops_of(T, (uint8_t*)P.ptr, P.pitch_bytes, P.height, div, x.rows, x.skip, ops, tap ? (const uint8_t*)G.ptr : nullptr, G.pitch_bytes, P.width))
            return fail(T, FAILURE, "staging buffer allocation failed");
    }
    return 0;
}

// the exchange plan of a dispatch list (rebuilt when the identifiers or a pass's reach change)
int build_plan(nrdhip_tiler& T, const uint32_t* ids, uint32_t n) {
    uint32_t count = 0;
    int r = nrdhip_dispatch_count(T.inst, ids, n, &count);
    if (r)
        return fail(T, r, std::string("dispatch list: ") + nrdhip_last_error(T.inst));
    std::vector<nrdhip_dispatch_info> d(count);
    // signature of the dispatch list: the ping-pong planes swap every frame and settings move a pass's reach, so the plan of the
    // previous frame is only reused when every dispatch reads / writes the same planes with the same reach
    std::vector<uint32_t> sig;
    for (uint32_t i = 0; i < count; i++) {
        if ((r = nrdhip_dispatch_info_get(T.inst, ids, n, i, &d[i])) != 0)
            return fail(T, r, "dispatch info");
        sig.push_back(0xffff0000u | d[i].halo_rows);
        sig.insert(sig.end(), d[i].written, d[i].written + d[i].written_num);
        sig.push_back(0xfffe0000u | (d[i].flags & 0xffffu));
        sig.insert(sig.end(), d[i].read, d[i].read + d[i].read_num);
        sig.push_back(0xfffd0000u);
        sig.insert(sig.end(), d[i].read_rows, d[i].read_rows + d[i].read_num);
        sig.push_back(0xfffc0000u);
        sig.insert(sig.end(), d[i].written_prefix, d[i].written_prefix + d[i].written_num);
        if (d[i].halo_rows > T.halo)
            return fail(T, INVALID, std::string(d[i].name) + " reads " + std::to_string(d[i].halo_rows) + " rows beyond its band, the band stores " +
                                        std::to_string(T.halo) + ": recreate the bands with nrdhip_required_halo() rows");
    }
    // the kernels must not trust previous-frame rows the plan never refreshes: the instance is told which local rows are current (owned
    // rows +- reprojRows; a footprint beyond them is rejected instead of being read from stale halo rows). Applied on EVERY call, cached
    // plan or not (the window is state of the instance - another caller, or a tiler before this one, may have moved it)
    auto apply_window = [&]() -> int {
        int32_t band[5];
        if (nrdhip_get_band(T.inst, band) != 0 || nrdhip_set_history_rows(T.inst, band[2] - (int32_t)T.reprojRows, (uint32_t)band[3] + 2u * T.reprojRows) != 0)
            return fail(T, FAILURE, "nrdhip_set_history_rows");
        return 0;
    };
    if (T.planIds == std::vector<uint32_t>(ids, ids + n) && T.planSig == sig)
        return apply_window();
    T.plan.assign(count, PlanEntry{});
    auto has = [](const uint32_t* v, uint32_t num, uint32_t code) { return std::find(v, v + num, code) != v + num; };
    // Rows of previous-frame state a band needs beyond its own: the motion allowance the stored halo leaves on top of the widest spatial
    // reach of the list (nrdhip_required_halo: reach + motion_rows, rounded up to 16) + 2 rows for the bilinear footprint - provided EVERY
    // read of previous-frame state (a permanent plane read before the list writes it) is a reprojected one or sits at the pixel's own
    // position; a single spatial reader of last frame's planes and the full halo travels (round 3's rule for every list).
    uint32_t reproj = 0, maxReach = 0;
    {
        bool provable = true;
        std::vector<uint32_t> writtenSoFar;
        T.firstPrevRead = count;
        for (uint32_t i = 0; i < count; i++) {
            maxReach = std::max<uint32_t>(maxReach, d[i].halo_rows);
            for (uint32_t k = 0; k < d[i].read_num; k++) {
                const uint32_t code = d[i].read[k];
                const bool previous = (code >> 16) == 0 && std::find(writtenSoFar.begin(), writtenSoFar.end(), code) == writtenSoFar.end();
                if (previous)
                    T.firstPrevRead = std::min(T.firstPrevRead, i);
                if (previous && d[i].read_rows[k] != 0 && d[i].read_rows[k] != NRDHIP_READ_REPROJECTED)
                    provable = false;
            }
            writtenSoFar.insert(writtenSoFar.end(), d[i].written, d[i].written + d[i].written_num);
        }
        reproj = provable ? std::min<uint32_t>(T.halo, T.halo - std::min(maxReach, T.halo) + 2u) : T.halo;
    }
    T.reprojRows = reproj;
    if ((r = apply_window()) != 0)
        return r;
    // how far dispatch j reads into plane `code` (read_rows: 0 = own pixel, N = spatial footprint, NRDHIP_READ_REPROJECTED)
    auto reach_into = [&](uint32_t j, uint32_t code) -> uint32_t {
        for (uint32_t k = 0; k < d[j].read_num; k++)
            if (d[j].read[k] == code)
                return d[j].read_rows[k] == NRDHIP_READ_REPROJECTED ? reproj : d[j].read_rows[k];
        return 0;
    };
    for (uint32_t i = 0; i < count; i++)
        for (uint32_t k = 0; k < d[i].written_num; k++) {
            const uint32_t code = d[i].written[k];
            if ((code >> 16) > 1 || (d[i].flags & NRDHIP_DISPATCH_ALL_ROWS))
                continue; // output slots are final; a pass over all stored rows (ClassifyTiles) leaves nothing to exchange
            uint32_t rows = 0;
            bool rewritten = false;
            for (uint32_t j = i + 1; j < count; j++) {
                if (has(d[j].read, d[j].read_num, code))
                    rows = std::max<uint32_t>(rows, reach_into(j, code));
                if (has(d[j].written, d[j].written_num, code)) {
                    rewritten = true;
                    break;
                }
            }
            // permanent planes that survive the frame: the next frame reprojects into them at motion-displaced rows (`reproj` rows, not the
            // whole halo), and nobody reads those rows before the next frame, so they travel deferred (minus what goes strips-first)
            if ((code >> 16) == 0 && !rewritten && rows < reproj)
                T.plan[i].later.push_back({code, reproj, rows, d[i].written_prefix[k]});
            if (rows > 0)
                T.plan[i].now.push_back({code, rows, 0, d[i].written_prefix[k]});
        }
    T.planIds.assign(ids, ids + n);
    T.planSig = sig;
    return 0;
}

int wait_deferred(nrdhip_tiler& T, nrdhip_tiler::DeferredSlot& slot, hipStream_t stream) {
    if (slot.pending) {
        if (!T.hostOrdered() && hipStreamWaitEvent(stream, slot.ev, 0) != hipSuccess)
            return fail(T, FAILURE, "hipStreamWaitEvent");
        slot.pending = false;
    }
    return 0;
}
// Every pending group of a list that shares a denoiser with `ids` (a denoiser may appear in two different lists - [REBLUR] one
// frame, [REBLUR, SIGMA] the next, or the same identifiers in another order - and ITS permanent rows are what must have arrived, whatever
// list sent them): the exact list is the common case, any intersection the rule
int wait_deferred_sharing(nrdhip_tiler& T, const uint32_t* ids, uint32_t n, hipStream_t stream) {
    for (auto& s : T.deferred) {
        if (!s.pending)
            continue;
        bool shares = false;
        for (uint32_t i = 0; i < n && !shares; i++)
            shares = std::find(s.ids.begin(), s.ids.end(), ids[i]) != s.ids.end();
        if (shares)
            if (int r = wait_deferred(T, s, stream))
                return r;
    }
    return 0;
}
nrdhip_tiler::DeferredSlot* deferred_slot(nrdhip_tiler& T, const uint32_t* ids, uint32_t n, hipStream_t stream) {
    for (auto& s : T.deferred)
        if (s.ids.size() == n && std::equal(s.ids.begin(), s.ids.end(), ids))
            return &s;
    if (T.deferred.size() >= 16) { // a caller that keeps inventing lists: drain and start over instead of growing one event per list for ever
        for (auto& s : T.deferred) {
            if (wait_deferred(T, s, stream))
                return nullptr;
            if (s.ev)
                (void)hipEventDestroy(s.ev);
        }
        T.deferred.clear();
    }
    nrdhip_tiler::DeferredSlot s;
    s.ids.assign(ids, ids + n);
    if (!T.hostOrdered() && hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess)
        return nullptr;
    T.deferred.push_back(s);
    return &T.deferred.back();
}

// the side stream the exchanges run on and the events that order it against the compute stream (RCCL and stream-ordered custom transports)
int make_side_stream(nrdhip_tiler& T) {
    if (hipStreamCreateWithFlags(&T.commStream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&T.evCompute, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&T.evComm, hipEventDisableTiming) != hipSuccess)
        return fail(T, FAILURE, "side stream / events");
    return 0;
}

// comm stream picks up after everything enqueued on the compute stream so far
int comm_after_compute(nrdhip_tiler& T, hipStream_t stream) {
    if (T.hostOrdered())
        return 0; // callbacks receive the compute stream and order themselves (host transports synchronise it)
    if (hipEventRecord(T.evCompute, stream) != hipSuccess || hipStreamWaitEvent(T.commStream, T.evCompute, 0) != hipSuccess)
        return fail(T, FAILURE, "hipEventRecord / hipStreamWaitEvent");
    return 0;
}

} // namespace

extern "C" {

NRDHIP_API int nrdhip_required_halo(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t motion_rows, uint32_t* out) {
    if (!inst || !out)
        return INVALID;
    uint32_t count = 0;
    int r = nrdhip_dispatch_count(inst, ids, n, &count);
    if (r)
        return r;
    uint32_t h = 0;
    for (uint32_t i = 0; i < count; i++) {
        nrdhip_dispatch_info d;
        if ((r = nrdhip_dispatch_info_get(inst, ids, n, i, &d)) != 0)
            return r;
        h = std::max<uint32_t>(h, d.halo_rows);
    }
    *out = (h + motion_rows + 15u) / 16u * 16u; // multiple of 16: band tile grids coincide with the single-GPU tile grid
    return 0;
}

NRDHIP_API int nrdhip_tiler_create(nrdhip_instance* inst, int rank, int world, const nrdhip_transport* transport, nrdhip_tiler** out) {
    if (!inst || !out || world < 1 || rank < 0 || rank >= world)
        return INVALID;
    int32_t band[5];
    if (nrdhip_get_band(inst, band) != 0)
        return INVALID;
    auto* T = new nrdhip_tiler();
    T->inst = inst;
    T->rank = rank;
    T->world = world;
    T->frameH = band[0];
    T->row0 = band[1];
    T->ownFirst = band[2];
    T->ownRows = band[3];
    T->localH = band[4];
    // rows stored beyond the owned band, on the side(s) that have a neighbour
    uint32_t above = (uint32_t)T->ownFirst, below = (uint32_t)(T->localH - T->ownFirst - T->ownRows);
    T->halo = world == 1 ? 0u : (rank == 0 ? below : (rank == world - 1 ? above : std::min(above, below)));
    if (world > 1 && (uint32_t)T->ownRows < T->halo) {
        delete T;
        return INVALID; // a band shorter than its halo cannot feed its neighbour's halo from owned rows
    }
    if (transport) {
        // (the struct is copied by value: a caller built against an older header or one that did not zero-initialise it would hand over
        // garbage flags - an unknown bit is refused rather than read as "stream-ordered"; nrdhip_sizeof(13) lets bindings check the layout)
        if (!transport->send || !transport->recv || (transport->flags & ~(uint32_t)NRDHIP_TRANSPORT_STREAM_ORDERED)) {
            delete T;
            return INVALID;
        }
        T->tr = *transport;
        T->custom = true;
        if (!T->hostOrdered()) {
            TilerDeviceScope scope(*T);
            if (make_side_stream(*T) != 0) {
                delete T;
                return FAILURE;
            }
        }
    }
    *out = T;
    return 0;
}

NRDHIP_API int nrdhip_tiler_rccl_unique_id(void* out128) {
    std::string err;
    if (!out128 || !g_rccl.load(err))
        return FAILURE;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    return g_rccl.GetUniqueId((ncclUniqueId*)out128) == ncclSuccess ? 0 : FAILURE;
}

NRDHIP_API int nrdhip_tiler_rccl_init(nrdhip_tiler* T, const void* unique_id128) {
    if (!T || !unique_id128 || T->custom)
        return INVALID;
    if (!g_rccl.load(T->error))
        return FAILURE;
    TilerDeviceScope scope(*T);
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&T->comm, T->world, id, T->rank); // on the instance's HIP device (one rank per GPU)
    if (r != ncclSuccess)
        return fail(*T, FAILURE, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    return make_side_stream(*T);
}

NRDHIP_API int nrdhip_tiler_rccl_loopback(nrdhip_tiler* T, const void* src, void* dst, size_t bytes, int deferred, void* stream) {
    if (!T || !src || !dst || !bytes || T->custom)
        return INVALID;
    if (!T->comm)
        return fail(*T, INVALID, "nrdhip_tiler_rccl_init has not been called");
    TilerDeviceScope scope(*T);
    hipStream_t st = (hipStream_t)stream;
    // [send, recv] toward one peer, the order ops_of gives every neighbour - the peer being this rank
    const std::vector<Op> ops = {Op{true, (uint8_t*)const_cast<void*>(src), bytes, T->rank}, Op{false, (uint8_t*)dst, bytes, T->rank}};
    int r = comm_after_compute(*T, st);
    if (!r)
        r = run_ops(*T, ops, T->commStream);
    if (r)
        return r;
    if (!deferred) { // an in-frame exchange: the compute stream goes on behind evComm
        if (hipEventRecord(T->evComm, T->commStream) != hipSuccess || hipStreamWaitEvent(st, T->evComm, 0) != hipSuccess)
            return fail(*T, FAILURE, "hipEventRecord / hipStreamWaitEvent");
        T->exchanges++;
        return 0;
    }
    // a deferred group: its own event, awaited where the list's next call would await it
    const uint32_t id = 0xffffffffu;
    nrdhip_tiler::DeferredSlot* slot = deferred_slot(*T, &id, 1, st);
    if (!slot)
        return fail(*T, FAILURE, "hipEventCreate");
    if (hipEventRecord(slot->ev, T->commStream) != hipSuccess)
        return fail(*T, FAILURE, "hipEventRecord");
    slot->pending = true;
    T->deferredExchanges++;
    return wait_deferred_sharing(*T, &id, 1, st);
}

NRDHIP_API void nrdhip_tiler_destroy(nrdhip_tiler* T) {
    if (!T)
        return;
    TilerDeviceScope scope(*T);
    if (T->world > 1 && nrdhip::instance_is_live(T->inst))
        (void)nrdhip_set_history_rows(T->inst, 0, 0); // an instance that outlives its tiler: every stored row is current again
    if (T->comm)
        g_rccl.CommDestroy(T->comm);
    if (T->commStream)
        (void)hipStreamDestroy(T->commStream);
    for (auto& slot : T->deferred)
        if (slot.ev)
            (void)hipEventDestroy(slot.ev);
    for (hipEvent_t e : {T->evCompute, T->evComm})
        if (e)
            (void)hipEventDestroy(e);
    for (auto& st : T->staging)
        (void)hipFree(st.ptr);
    delete T;
}

NRDHIP_API int nrdhip_tiler_halo(nrdhip_tiler* T, uint32_t* rows) {
    if (!T || !rows)
        return INVALID;
    *rows = T->halo;
    return 0;
}

// external inputs arrive per band (a renderer produces each band's rows): refresh the halo rows of the given bound slots
NRDHIP_API int nrdhip_tiler_exchange_inputs(nrdhip_tiler* T, const uint32_t* slots, uint32_t n, void* stream) {
    if (!T || (!slots && n))
        return INVALID;
    if (T->world == 1)
        return 0;
    TilerDeviceScope scope(*T);
    std::vector<Op> ops;
    for (uint32_t i = 0; i < n; i++) {
        nrdhip_plane_info P;
        if (nrdhip_slot_info(T->inst, slots[i], &P) != 0 || !P.ptr)
            return fail(*T, INVALID, "input slot not bound");
        if ((int32_t)P.height != T->localH)
            continue; // planes sampled by uv (confidence) are whole-frame, not banded
        ops_of(*T, (uint8_t*)P.ptr, P.pitch_bytes, P.height, 1, T->halo, 0, ops);
    }
    hipStream_t st = (hipStream_t)stream;
    int r = comm_after_compute(*T, st);
    if (!r)
        r = run_ops(*T, ops, T->hostOrdered() ? st : T->commStream);
    if (!r && !T->hostOrdered() && (hipEventRecord(T->evComm, T->commStream) != hipSuccess || hipStreamWaitEvent(st, T->evComm, 0) != hipSuccess))
        r = fail(*T, FAILURE, "hipEventRecord / hipStreamWaitEvent");
    T->exchanges++;
    return r;
}

NRDHIP_API int nrdhip_tiler_denoise(nrdhip_tiler* T, const uint32_t* ids, uint32_t n, void* stream) {
    if (!T || !ids || !n)
        return INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (T->world == 1)
        return nrdhip_denoise(T->inst, ids, n, stream);
    TilerDeviceScope scope(*T);
    int r = build_plan(*T, ids, n);
    if (r)
        return r;
    nrdhip_tiler::DeferredSlot* slot = deferred_slot(*T, ids, n, st);
    if (!slot)
        return fail(*T, FAILURE, "hipEventCreate");
    const bool up = T->rank > 0, down = T->rank < T->world - 1;
    std::vector<Xfer> laterAll;
    for (uint32_t i = 0; i < T->plan.size(); i++) {
        // the rows THIS list sent behind its previous call must have arrived before its first reader of previous-frame state runs
        // (the dispatches in front of it - ClassifyTiles - run while they are still travelling)
        if (i == T->firstPrevRead && (r = wait_deferred_sharing(*T, ids, n, st)) != 0)
            return r;
        const PlanEntry& e = T->plan[i];
        uint32_t strip = 0;
        for (auto& x : e.now)
            strip = std::max(strip, x.rows);
        strip = (strip + 15u) / 16u * 16u;
        std::vector<Op> ops;
        if ((r = collect(*T, e.now, ops)) != 0)
            return r;
        const bool split = !e.now.empty() && (up || down) && (uint32_t)T->ownRows >= 4 * strip && T->ownFirst % 16 == 0;
        if (!split) {
            if ((r = nrdhip_denoise_range(T->inst, ids, n, i, 1, stream)) != 0)
                return fail(*T, r, nrdhip_last_error(T->inst));
            if (!ops.empty()) {
                if ((r = comm_after_compute(*T, st)) != 0 || (r = run_ops(*T, ops, T->hostOrdered() ? st : T->commStream)) != 0)
                    return r;
                if (!T->hostOrdered() && (hipEventRecord(T->evComm, T->commStream) != hipSuccess || hipStreamWaitEvent(st, T->evComm, 0) != hipSuccess))
                    return fail(*T, FAILURE, "hipEventRecord / hipStreamWaitEvent");
                T->exchanges++;
            }
        } else {
            // boundary strips first, their rows travel while the interior is computed
            T->splitDispatches++;
            const uint32_t own0 = (uint32_t)T->ownFirst, ownN = (uint32_t)T->ownRows;
            const uint32_t lo = own0 + (up ? strip : 0), hi = own0 + ownN - (down ? strip : 0);
            bool first = true;
            if (up) {
                if ((r = nrdhip_denoise_rows(T->inst, ids, n, i, own0, strip, NRDHIP_PART_FIRST, stream)) != 0)
                    return fail(*T, r, nrdhip_last_error(T->inst));
                first = false;
            }
            if (down && (r = nrdhip_denoise_rows(T->inst, ids, n, i, hi, own0 + ownN - hi, first ? NRDHIP_PART_FIRST : 0u, stream)) != 0)
                return fail(*T, r, nrdhip_last_error(T->inst));
            if ((r = comm_after_compute(*T, st)) != 0 || (r = run_ops(*T, ops, T->hostOrdered() ? st : T->commStream)) != 0)
                return r;
            if (!T->hostOrdered() && hipEventRecord(T->evComm, T->commStream) != hipSuccess)
                return fail(*T, FAILURE, "hipEventRecord");
            if ((r = nrdhip_denoise_rows(T->inst, ids, n, i, lo, hi - lo, NRDHIP_PART_LAST, stream)) != 0)
                return fail(*T, r, nrdhip_last_error(T->inst));
            if (!T->hostOrdered() && hipStreamWaitEvent(st, T->evComm, 0) != hipSuccess)
                return fail(*T, FAILURE, "hipStreamWaitEvent");
            T->exchanges++;
        }
        laterAll.insert(laterAll.end(), e.later.begin(), e.later.end());
    }
    if (T->firstPrevRead >= T->plan.size() && (r = wait_deferred_sharing(*T, ids, n, st)) != 0) // (a list without a reader of previous-frame state)
        return r;
    if (!laterAll.empty()) { // ONE group behind the last dispatch: nothing of it is read before this list's next call
        std::vector<Op> lops;
        if ((r = collect(*T, laterAll, lops)) != 0)
            return r;
        if (!lops.empty()) {
            if ((r = comm_after_compute(*T, st)) != 0 || (r = run_ops(*T, lops, T->hostOrdered() ? st : T->commStream)) != 0)
                return r;
            if (!T->hostOrdered() && hipEventRecord(slot->ev, T->commStream) != hipSuccess)
                return fail(*T, FAILURE, "hipEventRecord");
            slot->pending = true;
            T->deferredExchanges++;
        }
    }
    return 0;
}

// rows only the next frame reads may still be travelling: make `stream` wait for them (end of a run, before reading pool planes)
NRDHIP_API int nrdhip_tiler_finish(nrdhip_tiler* T, void* stream) {
    if (!T)
        return INVALID;
    TilerDeviceScope scope(*T);
    for (auto& slot : T->deferred)
        if (int r = wait_deferred(*T, slot, (hipStream_t)stream))
            return r;
    return 0;
}

NRDHIP_API int nrdhip_tiler_stats(nrdhip_tiler* T, uint64_t out[4]) {
    if (!T || !out)
        return INVALID;
    out[0] = T->bytesSent;
    out[1] = T->splitDispatches;
    out[2] = T->exchanges;
    out[3] = T->deferredExchanges;
    return 0;
}

NRDHIP_API const char* nrdhip_tiler_last_error(nrdhip_tiler* T) { return T ? T->error.c_str() : "null tiler"; }
}
