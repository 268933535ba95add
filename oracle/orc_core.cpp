// orc_core.cpp - CPU ORACLE: instance, pools, constants, C entry points (orc_*).
// Test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the product
// (libnrdhip.so) never links or loads this. PARITY UNPINNED vs upstream NRD (see oracle/README.md).
//
// The entry points mirror include/nrdhip.h one to one (prefix orc_ instead of nrdhip_) so the same
// host-side driver code can run either backend; "device pointers" are host pointers here and the stream
// argument is ignored.
#include "orc_core.h"
#include "../include/nrdhip.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <unistd.h>

namespace orc {

const float g_poisson8[8][3] = {
    {-0.4706069f, -0.4427112f, 0.7592f}, {-0.9057375f, 0.3003471f, 0.5483f}, {-0.3487388f, 0.4037880f, 0.8287f},
    {0.1023042f, 0.6439373f, 0.7554f},   {0.5699277f, 0.3513750f, 0.7439f},  {0.2939128f, -0.1131226f, 0.9366f},
    {0.7836658f, -0.4208784f, 0.5932f},  {0.1564120f, -0.8198990f, 0.6314f}};

// CommonSettings -> per-frame constants. Column-major 4x4 input (NRDSample.cpp:3836-3839).
bool derive_consts(const nrd::CommonSettings& cs, int resW, int resH, int frameH, int yOff, int ownY0, int ownRows, Consts& c, std::string& err, int histY0, int histRows) {
    c = Consts();
    c.W = cs.rectSize[0];
    c.H = cs.rectSize[1];
    c.Wprev = cs.rectSizePrev[0] ? cs.rectSizePrev[0] : c.W;
    c.Hprev = cs.rectSizePrev[1] ? cs.rectSizePrev[1] : c.H;
    c.resW = resW;
    c.resH = resH;
    c.yOff = yOff;
    if (c.W <= 0 || c.H <= 0 || c.W > resW) {
        err = "rectSize invalid";
        return false;
    }
    if (frameH == resH && yOff == 0) {
        if (c.H > resH) {
            err = "rectSize exceeds resourceSize";
            return false;
        }
    }
    // owned local rows, clipped to the rect
    c.ownY0 = ownY0;
    c.ownY1 = ownRows ? ownY0 + ownRows : resH;
    c.ownY0 = std::max(c.ownY0, -yOff);
    c.ownY1 = std::min(c.ownY1, c.H - yOff);
    c.ownY1 = std::min(c.ownY1, resH);
    if (c.ownY1 < c.ownY0)
        c.ownY1 = c.ownY0;
    c.prevY0 = histRows ? std::max(histY0, 0) : 0;
    c.prevY1 = histRows ? std::min(histY0 + histRows, resH) : resH;
    c.invW = 1.0f / (float)c.W;
    c.invH = 1.0f / (float)c.H;
    c.invWprev = 1.0f / (float)c.Wprev;
    c.invHprev = 1.0f / (float)c.Hprev;

    // cameraJitter (pixels, Source/NRDSample.cpp:3843-3846): the G-buffer of pixel (x, y) was rendered through uv + jitter / rect
    // (Shaders/Composition.cs.hlsl:77 "pixelUv + gJitter") while the matrices are un-jittered - fold the constant uv offset into
    // the projection's x/y shear terms so that reconstruct / project / the tap Jacobian all see the jittered pixel grid
    auto is_ortho = [](const float* M) { return M[11] == 0.0f && M[15] != 0.0f; };
    c.ortho = is_ortho(cs.viewToClipMatrix);
    const float orthoFlag = c.ortho ? 1.0f : 0.0f;
    auto proj = [&](const float* M, float* pj, float* fr, const float* jitter, float invW, float invH) -> bool {
        fr[4] = orthoFlag;
        pj[5] = orthoFlag;
        if (c.ortho) { // clip.w == m15: the constant uv offset of the jitter goes into m12 / m13
            float w = M[15];
            float m0 = M[0] / w, m5 = M[5] / w;
            float m12 = M[12] / w - 2.0f * jitter[0] * invW, m13 = M[13] / w + 2.0f * jitter[1] * invH;
            if (m0 == 0.0f || m5 == 0.0f)
                return false;
            pj[0] = m0;
            pj[1] = m5;
            pj[2] = m12;
            pj[3] = m13;
            pj[4] = 1.0f;
            fr[2] = 2.0f / m0;
            fr[0] = (-1.0f - m12) / m0;
            fr[3] = -2.0f / m5;
            fr[1] = (1.0f - m13) / m5;
            return true;
        }
        float s = M[11];
        if (s == 0.0f)
            return false;
        s = s > 0.0f ? 1.0f : -1.0f;
        float m0 = M[0], m5 = M[5];
        float m8 = M[8] - 2.0f * s * jitter[0] * invW, m9 = M[9] + 2.0f * s * jitter[1] * invH;
        if (m0 == 0.0f || m5 == 0.0f)
            return false;
        pj[0] = m0;
        pj[1] = m5;
        pj[2] = m8;
        pj[3] = m9;
        pj[4] = s;
        fr[2] = 2.0f * s / m0;
        fr[0] = (-s - m8) / m0;
        fr[3] = -2.0f * s / m5;
        fr[1] = (s - m9) / m5;
        return true;
    };
    // a projection-mode switch comes with an accumulation restart (Source/NRDSample.cpp:2142): the current matrix stands in for the previous one
    const float* Mprev = is_ortho(cs.viewToClipMatrixPrev) == c.ortho ? cs.viewToClipMatrixPrev : cs.viewToClipMatrix;
    if (!proj(cs.viewToClipMatrix, c.pj, c.fr, cs.cameraJitter, c.invW, c.invH) || !proj(Mprev, c.pjPrev, c.frPrev, cs.cameraJitterPrev, c.invWprev, c.invHprev)) {
        err = "viewToClipMatrix is neither a perspective nor an orthographic projection";
        return false;
    }
    for (int k = 0; k < 2; k++) {
        float* pv = k ? c.pvPrev : c.pv;
        const float* fr = k ? c.frPrev : c.fr;
        float iw = k ? c.invWprev : c.invW, ih = k ? c.invHprev : c.invH;
        pv[2] = fr[2] * iw;
        pv[3] = fr[3] * ih;
        pv[0] = fr[0] + 0.5f * pv[2];
        pv[1] = fr[1] + 0.5f * pv[3];
        pv[4] = orthoFlag;
    }
    auto rotpos = [&](const float* M, float* R, float* pos) {
        for (int r = 0; r < 3; r++)
            for (int col = 0; col < 3; col++)
                R[r * 3 + col] = M[col * 4 + r];
        float t[3] = {M[12], M[13], M[14]};
        for (int i = 0; i < 3; i++) // pos = -R^T t
            pos[i] = -(R[0 * 3 + i] * t[0] + R[1 * 3 + i] * t[1] + R[2 * 3 + i] * t[2]);
    };
    float pos[3], posPrev[3];
    rotpos(cs.worldToViewMatrix, c.w2v, pos);
    rotpos(cs.worldToViewMatrixPrev, c.w2vPrev, posPrev);
    for (int r = 0; r < 3; r++)
        for (int col = 0; col < 3; col++)
            c.v2w[r * 3 + col] = c.w2v[col * 3 + r], c.v2wPrev[r * 3 + col] = c.w2vPrev[col * 3 + r];
    for (int i = 0; i < 3; i++)
        c.camDelta[i] = posPrev[i] - pos[i];

    c.unproject = 1.0f / (0.5f * (float)c.H * absf(c.pj[1]));
    c.minRectDimMulUnproject = (float)std::min(c.W, c.H) * c.unproject;
    c.jcx = 0.5f * (float)c.W * c.pj[0] * c.unproject * c.pj[4]; // (pj[4] = +-1)
    c.jcy = -0.5f * (float)c.H * c.pj[1] * c.unproject * c.pj[4];
    c.denoisingRange = cs.denoisingRange;
    c.disocclusionThreshold = cs.disocclusionThreshold;
    c.disoccAlt = cs.disocclusionThresholdAlternate;
    c.mixAvail = cs.isDisocclusionThresholdMixAvailable;
    c.splitScreen = cs.splitScreen;
    for (int i = 0; i < 3; i++)
        c.mvScale[i] = cs.motionVectorScale[i];
    c.frameIndex = cs.frameIndex;
    c.strandMat = (cs.strandMaterialID >= 0.0f && cs.strandMaterialID <= 3.0f) ? (uint32_t)cs.strandMaterialID : 0xffffffffu;
    c.strandThickness = cs.strandThickness;
    c.camAttachMat = (cs.cameraAttachedReflectionMaterialID >= 0.0f && cs.cameraAttachedReflectionMaterialID <= 3.0f) ? (uint32_t)cs.cameraAttachedReflectionMaterialID : 0xffffffffu;
    c.mvWorld = cs.isMotionVectorInWorldSpace;
    c.confAvail = cs.isHistoryConfidenceAvailable;
    c.reset = cs.accumulationMode != nrd::AccumulationMode::CONTINUE;
    for (int k = 0; k < 64; k++) {
        double a = 6.283185307179586 * (double)k / 64.0;
        c.rot[k][0] = (float)std::cos(a);
        c.rot[k][1] = (float)std::sin(a);
    }
    return true;
}

static DenoiserState* find(Instance& I, uint32_t id) {
    for (auto& d : I.denoisers)
        if (d.identifier == id)
            return &d;
    return nullptr;
}

static void describe(DenoiserState& d, std::vector<PoolPlane>& perm, std::vector<PoolPlane>& trans) {
    switch (d.kind) {
        case Kind::REFERENCE: reference_describe(d, perm, trans); break;
        case Kind::REBLUR: reblur_describe(d, perm, trans); break;
        case Kind::SIGMA: sigma_describe(d, perm, trans); break;
        case Kind::RELAX: relax_describe(d, perm, trans); break;
    }
}

static void build(Instance& I, DenoiserState& d) {
    d.passes.clear();
    switch (d.kind) {
        case Kind::REFERENCE: reference_build(I, d); break;
        case Kind::REBLUR: reblur_build(I, d); break;
        case Kind::SIGMA: sigma_build(I, d); break;
        case Kind::RELAX: relax_build(I, d); break;
    }
}

static bool classify(nrd::Denoiser dn, DenoiserState& d) {
    using D = nrd::Denoiser;
    d.denoiser = dn;
    switch (dn) {
        case D::REBLUR_DIFFUSE: d.kind = Kind::REBLUR; d.hasDiff = true; break;
        case D::REBLUR_SPECULAR: d.kind = Kind::REBLUR; d.hasSpec = true; break;
        case D::REBLUR_DIFFUSE_SPECULAR: d.kind = Kind::REBLUR; d.hasDiff = d.hasSpec = true; break;
        case D::REBLUR_DIFFUSE_OCCLUSION: d.kind = Kind::REBLUR; d.hasDiff = d.occlusion = true; break;
        case D::REBLUR_SPECULAR_OCCLUSION: d.kind = Kind::REBLUR; d.hasSpec = d.occlusion = true; break;
        case D::REBLUR_DIFFUSE_SPECULAR_OCCLUSION: d.kind = Kind::REBLUR; d.hasDiff = d.hasSpec = d.occlusion = true; break;
        case D::REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION: d.kind = Kind::REBLUR; d.hasDiff = d.sh = d.dirOcc = true; break;
        case D::REBLUR_DIFFUSE_SH: d.kind = Kind::REBLUR; d.hasDiff = d.sh = true; break;
        case D::REBLUR_SPECULAR_SH: d.kind = Kind::REBLUR; d.hasSpec = d.sh = true; break;
        case D::REBLUR_DIFFUSE_SPECULAR_SH: d.kind = Kind::REBLUR; d.hasDiff = d.hasSpec = d.sh = true; break;
        case D::RELAX_DIFFUSE_SH: d.kind = Kind::RELAX; d.hasDiff = d.sh = true; break;
        case D::RELAX_SPECULAR_SH: d.kind = Kind::RELAX; d.hasSpec = d.sh = true; break;
        case D::RELAX_DIFFUSE_SPECULAR_SH: d.kind = Kind::RELAX; d.hasDiff = d.hasSpec = d.sh = true; break;
        case D::RELAX_DIFFUSE: d.kind = Kind::RELAX; d.hasDiff = true; break;
        case D::RELAX_SPECULAR: d.kind = Kind::RELAX; d.hasSpec = true; break;
        case D::RELAX_DIFFUSE_SPECULAR: d.kind = Kind::RELAX; d.hasDiff = d.hasSpec = true; break;
        case D::SIGMA_SHADOW: d.kind = Kind::SIGMA; break;
        case D::SIGMA_SHADOW_TRANSLUCENCY: d.kind = Kind::SIGMA; d.translucency = true; break;
        case D::REFERENCE: d.kind = Kind::REFERENCE; break;
        default: return false;
    }
    d.nsig = (d.hasDiff ? 1 : 0) + (d.hasSpec ? 1 : 0);
    return true;
}

// ---- persistent worker pool of the oracle (process-wide, grows to the largest thread count asked for, never shrinks) -------------
namespace {
struct Pool {
    std::mutex m;
    std::condition_variable wake, done;
    std::vector<std::thread>* workers = new std::vector<std::thread>(); // (heap: a forked child abandons the parent's handles, below)
    pid_t owner = getpid();
    const std::function<void(int)>* job = nullptr;
    std::atomic<int> next{0};
    int tasks = 0, active = 0, allowed = 0;
    uint64_t generation = 0;
    bool quit = false;
    std::mutex callers; // one parallel region at a time (instances on several host threads take turns)

    void work(const std::function<void(int)>& f, int nTasks) {
        for (int k = next.fetch_add(1, std::memory_order_relaxed); k < nTasks; k = next.fetch_add(1, std::memory_order_relaxed))
            f(k);
    }
    void loop(int index) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            wake.wait(lk, [&] { return quit || (generation != seen && index < allowed); });
            if (quit)
                return;
            seen = generation;
            const std::function<void(int)>* f = job;
            const int nTasks = tasks;
            lk.unlock();
            work(*f, nTasks);
            lk.lock();
            if (--active == 0)
                done.notify_one();
        }
    }
    void run(int threads, int nTasks, const std::function<void(int)>& f) {
        std::lock_guard<std::mutex> one(callers);
        const int helpers = std::min(threads, nTasks) - 1; // the caller works too
        {
            std::unique_lock<std::mutex> lk(m);
            if (owner != getpid()) { // a forked child: the parent's workers do not exist here - start over (their handles are left alone)
                workers = new std::vector<std::thread>();
                owner = getpid();
            }
            while ((int)workers->size() < helpers) {
                const int index = (int)workers->size();
                workers->emplace_back([this, index] { loop(index); });
            }
            job = &f;
            tasks = nTasks;
            next.store(0, std::memory_order_relaxed);
            active = helpers;
            allowed = helpers;
            generation++;
        }
        wake.notify_all();
        work(f, nTasks);
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return active == 0; });
        job = nullptr;
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        wake.notify_all();
        if (owner == getpid())
            for (auto& t : *workers)
                t.join();
    }
};
Pool& pool() {
    static Pool P;
    return P;
}
void pool_run(int threads, int nTasks, const std::function<void(int)>& f) { pool().run(threads, nTasks, f); }
} // namespace

static void run_pass(Instance& I, DenoiserState& d, const Consts& c, Pass& p) {
    int y0 = c.ownY0, y1 = c.ownY1;
    if (p.allRows) { // every row the band stores (csrc/nrdhip.cpp on_all_rows)
        y0 = std::max(0, -c.yOff);
        y1 = std::max(y0, std::min(c.resH, c.H - c.yOff));
    }
    if (p.tileGrid) { // iterate whole tiles covering the owned rows
        y0 = y0 / 16;
        y1 = (y1 + 15) / 16;
    }
    int n = std::max(1, std::min(I.threads, y1 - y0));
    if (n == 1) {
        p.run(I, d, c, y0, y1);
        return;
    }
    // Rows in small chunks handed out through one atomic counter to persistent workers (round 6: the baseline bench.py reports used to
    // spawn `threads` std::threads per pass over a STATIC split - a frame whose upper third is sky kept a third of them idle, and 256
    // spawns x 7 passes per frame cost more than the passes of a small frame). A chunk is a few rows: every pass writes rows of its own
    // range only, so any partition of [y0, y1) gives the same planes.
    const int rows = y1 - y0;
    const int chunk = std::max(1, rows / (n * 8));
    pool_run(n, (rows + chunk - 1) / chunk, [&](int k) {
        const int a = y0 + k * chunk, b = std::min(y1, a + chunk);
        p.run(I, d, c, a, b);
    });
}

struct Flat {
    DenoiserState* d;
    uint32_t passIndex;
};

static int flatten(Instance& I, const uint32_t* ids, uint32_t n, std::vector<Flat>& out) {
    out.clear();
    for (uint32_t i = 0; i < n; i++) {
        DenoiserState* d = find(I, ids[i]);
        if (!d) {
            I.error = "unknown identifier";
            return (int)nrd::Result::INVALID_ARGUMENT;
        }
        build(I, *d);
        for (uint32_t k = 0; k < d->passes.size(); k++)
            out.push_back({d, k});
    }
    return 0;
}

} // namespace orc

using namespace orc;

struct nrdhip_instance {
    Instance I;
};

static std::string g_createError;

extern "C" {

NRDHIP_API int orc_create(const nrdhip_create_desc* desc, nrdhip_instance** out) {
    if (!desc || !out || !desc->denoisers || !desc->denoisers_num || !desc->resource_width || !desc->resource_height) {
        g_createError = "invalid creation desc";
        return (int)nrd::Result::INVALID_ARGUMENT;
    }
    auto* h = new nrdhip_instance();
    Instance& I = h->I;
    I.resW = desc->resource_width;
    I.resH = desc->resource_height;
    I.frameH = desc->frame_height ? desc->frame_height : desc->resource_height;
    I.yOff = desc->band_row0;
    I.ownY0 = desc->band_own_first;
    I.ownRows = desc->band_own_rows;
    I.flags = desc->flags;
    for (uint32_t i = 0; i < desc->denoisers_num; i++) {
        DenoiserState d;
        d.identifier = desc->denoisers[i].identifier;
        if (find(I, d.identifier)) {
            g_createError = "non unique identifier";
            delete h;
            return (int)nrd::Result::NON_UNIQUE_IDENTIFIER;
        }
        if (desc->denoisers[i].denoiser >= (uint32_t)nrd::Denoiser::MAX_NUM || !classify((nrd::Denoiser)desc->denoisers[i].denoiser, d)) {
            g_createError = "unsupported denoiser";
            delete h;
            return (int)nrd::Result::UNSUPPORTED;
        }
        d.permBase = (uint32_t)I.permDesc.size();
        d.transBase = (uint32_t)I.transDesc.size();
        describe(d, I.permDesc, I.transDesc);
        I.denoisers.push_back(d);
    }
    auto alloc = [&](std::vector<PoolPlane>& descs, std::vector<Plane>& planes) {
        for (auto& pd : descs) {
            Plane P;
            P.fmt = pd.fmt;
            P.bpt = pd.bpt;
            P.name = pd.name;
            P.w = (uint16_t)((I.resW + pd.downsample - 1) / pd.downsample);
            P.h = (uint16_t)((I.resH + pd.downsample - 1) / pd.downsample);
            P.pitch = P.w * P.bpt;
            if (!(I.flags & NRDHIP_FLAG_EXTERNAL_POOLS)) {
                I.owned.emplace_back((size_t)P.pitch * P.h, 0);
                P.p = I.owned.back().data();
            }
            planes.push_back(P);
        }
    };
    alloc(I.permDesc, I.perm);
    alloc(I.transDesc, I.trans);
    *out = h;
    return 0;
}

NRDHIP_API void orc_destroy(nrdhip_instance* inst) { delete inst; }
NRDHIP_API int orc_new_frame(nrdhip_instance*) { return 0; }
NRDHIP_API int orc_set_history_rows(nrdhip_instance* inst, int32_t first_local_row, uint32_t rows) {
    if (!inst)
        return (int)nrd::Result::INVALID_ARGUMENT;
    inst->I.histY0 = first_local_row;
    inst->I.histRows = (int)rows;
    return 0;
}

NRDHIP_API int orc_set_threads(nrdhip_instance* inst, int threads) {
    inst->I.threads = threads < 1 ? 1 : threads;
    return 0;
}

NRDHIP_API int orc_set_common(nrdhip_instance* inst, const void* settings, size_t size) {
    if (!inst || !settings || size != sizeof(nrd::CommonSettings))
        return (int)nrd::Result::INVALID_ARGUMENT;
    std::memcpy(&inst->I.common, settings, size);
    inst->I.commonSet = true;
    return 0;
}

NRDHIP_API int orc_set_denoiser(nrdhip_instance* inst, uint32_t identifier, const void* settings, size_t size) {
    if (!inst || !settings)
        return (int)nrd::Result::INVALID_ARGUMENT;
    DenoiserState* d = find(inst->I, identifier);
    if (!d)
        return (int)nrd::Result::INVALID_ARGUMENT;
    void* dst = nullptr;
    size_t want = 0;
    switch (d->kind) {
        case Kind::REBLUR: dst = &d->reblur; want = sizeof(d->reblur); break;
        case Kind::RELAX: dst = &d->relax; want = sizeof(d->relax); break;
        case Kind::SIGMA: dst = &d->sigma; want = sizeof(d->sigma); break;
        case Kind::REFERENCE: dst = &d->reference; want = sizeof(d->reference); break;
    }
    if (size != want)
        return (int)nrd::Result::INVALID_ARGUMENT;
    std::memcpy(dst, settings, size);
    return 0;
}

NRDHIP_API int orc_bind(nrdhip_instance* inst, uint32_t slot, void* ptr, uint32_t pitch, uint32_t format, uint16_t w, uint16_t h) {
    if (!inst || slot >= (uint32_t)nrd::ResourceType::TRANSIENT_POOL)
        return (int)nrd::Result::INVALID_ARGUMENT;
    Plane& P = inst->I.slots[slot];
    P.p = (uint8_t*)ptr;
    P.pitch = pitch;
    P.fmt = format;
    P.w = w;
    P.h = h;
    P.bpt = format_bytes(format);
    return 0;
}

NRDHIP_API int orc_pool_size(nrdhip_instance* inst, uint32_t pool, uint32_t* count) {
    *count = (uint32_t)(pool == 0 ? inst->I.perm.size() : inst->I.trans.size());
    return 0;
}

NRDHIP_API int orc_pool_info(nrdhip_instance* inst, uint32_t pool, uint32_t index, nrdhip_plane_info* out) {
    auto& v = pool == 0 ? inst->I.perm : inst->I.trans;
    if (index >= v.size())
        return (int)nrd::Result::INVALID_ARGUMENT;
    const Plane& P = v[index];
    out->ptr = P.p;
    out->pitch_bytes = P.pitch;
    out->format = P.fmt;
    out->width = P.w;
    out->height = P.h;
    out->bytes_per_texel = P.bpt;
    out->name = P.name;
    return 0;
}

NRDHIP_API int orc_bind_pool(nrdhip_instance* inst, uint32_t pool, uint32_t index, void* ptr, uint32_t pitch) {
    auto& v = pool == 0 ? inst->I.perm : inst->I.trans;
    if (index >= v.size() || pitch < v[index].w * v[index].bpt)
        return (int)nrd::Result::INVALID_ARGUMENT;
    v[index].p = (uint8_t*)ptr;
    v[index].pitch = pitch;
    return 0;
}

NRDHIP_API int orc_dispatch_count(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t* count) {
    std::vector<Flat> fl;
    int r = flatten(inst->I, ids, n, fl);
    *count = (uint32_t)fl.size();
    return r;
}

NRDHIP_API int orc_dispatch_info_get(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t index, nrdhip_dispatch_info* out) {
    std::vector<Flat> fl;
    int r = flatten(inst->I, ids, n, fl);
    if (r)
        return r;
    if (index >= fl.size())
        return (int)nrd::Result::INVALID_ARGUMENT;
    const Pass& p = fl[index].d->passes[fl[index].passIndex];
    std::memset(out, 0, sizeof(*out));
    out->name = p.name;
    out->kernel = p.kernel;
    out->identifier = fl[index].d->identifier;
    int div = 16;
    out->grid_width = (uint16_t)((inst->I.common.rectSize[0] + div - 1) / div);
    out->grid_height = (uint16_t)((inst->I.common.rectSize[1] + div - 1) / div);
    out->halo_rows = p.haloRows;
    out->written_num = (uint16_t)std::min<size_t>(p.written.size(), 12);
    for (uint32_t i = 0; i < out->written_num; i++)
        out->written[i] = p.written[i];
    out->read_num = (uint32_t)std::min<size_t>(p.read.size(), 24);
    for (uint32_t i = 0; i < out->read_num; i++) {
        out->read[i] = p.read[i];
        uint16_t rows = p.haloRows;
        if (std::find(p.own.begin(), p.own.end(), p.read[i]) != p.own.end())
            rows = 0;
        if (std::find(p.reprojected.begin(), p.reprojected.end(), p.read[i]) != p.reprojected.end())
            rows = (uint16_t)NRDHIP_READ_REPROJECTED;
        for (auto& rc : p.reach)
            if (rc.first == p.read[i])
                rows = std::min(rc.second, p.haloRows);
        out->read_rows[i] = rows;
    }
    out->flags = p.allRows ? (uint32_t)NRDHIP_DISPATCH_ALL_ROWS : 0u;
    for (uint32_t i = 0; i < 12; i++) {
        out->written_prefix[i] = (uint32_t)NRDHIP_NO_PLANE;
        for (auto& pf : p.prefix)
            if (i < out->written_num && pf.first == out->written[i])
                out->written_prefix[i] = pf.second;
    }
    out->algorithmic_bytes_per_pixel = p.bytesPerPixel;
    return 0;
}

static int denoise_parts(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t first, uint32_t count, uint32_t part);

NRDHIP_API int orc_denoise_range(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t first, uint32_t count, void*) {
    return denoise_parts(inst, ids, n, first, count, NRDHIP_PART_FIRST | NRDHIP_PART_LAST);
}

// mirror of nrdhip_denoise_rows (include/nrdhip.h): one pass restricted to a window of the owned local rows
NRDHIP_API int orc_denoise_rows(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t index, uint32_t row_first, uint32_t row_count,
                                uint32_t part, void*) {
    if (!inst)
        return (int)nrd::Result::INVALID_ARGUMENT;
    Instance& I = inst->I;
    const int own0 = I.ownY0, ownN = I.ownRows;
    int lo = std::max((int)row_first, own0), hi = (int)(row_first + row_count);
    if (ownN)
        hi = std::min(hi, own0 + ownN);
    hi = std::min(hi, (int)I.resH);
    int r = 0;
    if (hi > lo) {
        I.ownY0 = lo;
        I.ownRows = hi - lo;
        r = denoise_parts(inst, ids, n, index, 1, part);
        I.ownY0 = own0;
        I.ownRows = ownN;
    } else if (part & NRDHIP_PART_LAST)
        r = denoise_parts(inst, ids, n, index, 1, (part & ~NRDHIP_PART_FIRST) | 4u);
    return r;
}

static int denoise_parts(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, uint32_t first, uint32_t count, uint32_t part) {
    if (!inst || !inst->I.commonSet)
        return (int)nrd::Result::INVALID_ARGUMENT;
    Instance& I = inst->I;
    for (auto& P : I.perm)
        if (!P.p) {
            I.error = "pool plane not bound";
            return (int)nrd::Result::INVALID_ARGUMENT;
        }
    for (auto& P : I.trans)
        if (!P.p) {
            I.error = "pool plane not bound";
            return (int)nrd::Result::INVALID_ARGUMENT;
        }
    Consts c;
    if (!derive_consts(I.common, I.resW, I.resH, I.frameH, I.yOff, I.ownY0, I.ownRows, c, I.error, I.histY0, I.histRows))
        return (int)nrd::Result::INVALID_ARGUMENT;
    std::vector<Flat> fl;
    int r = flatten(I, ids, n, fl);
    if (r)
        return r;
    if (first > fl.size() || first + count > fl.size())
        return (int)nrd::Result::INVALID_ARGUMENT;
    for (uint32_t i = first; i < first + count; i++) {
        DenoiserState& d = *fl[i].d;
        Pass& p = d.passes[fl[i].passIndex];
        for (uint32_t s : p.read)
            if ((s >> 16) == 2 && !I.slots[s & 0xffff].p) {
                I.error = std::string("resource slot not bound for pass ") + p.name;
                return (int)nrd::Result::INVALID_ARGUMENT;
            }
        for (const auto* lst : {&p.read, &p.written})
            for (uint32_t s : *lst)
                if ((s >> 16) == 2 && I.slots[s & 0xffff].p && !nrd::IsFormatAllowed((nrd::ResourceType)(s & 0xffff), (nrd::Format)I.slots[s & 0xffff].fmt)) {
                    I.error = std::string("resource slot bound with an unsupported format for pass ") + p.name;
                    return (int)nrd::Result::INVALID_ARGUMENT;
                }
        for (uint32_t s : p.written)
            if ((s >> 16) == 2 && !I.slots[s & 0xffff].p) {
                I.error = std::string("output slot not bound for pass ") + p.name;
                return (int)nrd::Result::INVALID_ARGUMENT;
            }
        if ((part & NRDHIP_PART_FIRST) && fl[i].passIndex == 0 && (I.common.accumulationMode == nrd::AccumulationMode::CLEAR_AND_RESTART || !d.historyValid)) {
            // clear this denoiser's permanent planes (history content is discarded)
            uint32_t end = (uint32_t)I.perm.size();
            for (auto& o : I.denoisers)
                if (o.permBase > d.permBase)
                    end = std::min(end, o.permBase);
            for (uint32_t k = d.permBase; k < end; k++)
                for (int y = 0; y < I.perm[k].h; y++)
                    std::memset(I.perm[k].p + (size_t)y * I.perm[k].pitch, 0, (size_t)I.perm[k].w * I.perm[k].bpt);
        }
        if (!(part & 4u))
            run_pass(I, d, c, p);
        if ((part & NRDHIP_PART_LAST) && fl[i].passIndex + 1 == d.passes.size()) { // frame of this denoiser complete
            d.framesSinceReset = (c.reset || !d.historyValid) ? 1 : d.framesSinceReset + 1;
            d.frameCounter++;
            d.historyValid = true;
        }
    }
    return 0;
}

NRDHIP_API int orc_denoise(nrdhip_instance* inst, const uint32_t* ids, uint32_t n, void* stream) {
    uint32_t count = 0;
    int r = orc_dispatch_count(inst, ids, n, &count);
    if (r)
        return r;
    return orc_denoise_range(inst, ids, n, 0, count, stream);
}

NRDHIP_API int orc_get_memory_mb(nrdhip_instance* inst, float out[3]) {
    double p = 0, t = 0;
    for (auto& P : inst->I.perm)
        p += (double)P.w * P.h * P.bpt;
    for (auto& P : inst->I.trans)
        t += (double)P.w * P.h * P.bpt;
    out[0] = (float)((p + t) / 1048576.0);
    out[1] = (float)(p / 1048576.0);
    out[2] = (float)(t / 1048576.0);
    return 0;
}

// checkpoint / resume counters (include/nrdhip.h nrdhip_history_state): what a denoiser carries between frames besides its permanent planes
NRDHIP_API int orc_get_history_state(nrdhip_instance* inst, uint32_t identifier, nrdhip_history_state* out) {
    DenoiserState* d = inst ? find(inst->I, identifier) : nullptr;
    if (!d || !out)
        return (int)nrd::Result::INVALID_ARGUMENT;
    *out = {d->frameCounter, d->framesSinceReset, d->historyValid ? 1u : 0u, 0u};
    return 0;
}
NRDHIP_API int orc_set_history_state(nrdhip_instance* inst, uint32_t identifier, const nrdhip_history_state* in) {
    DenoiserState* d = inst ? find(inst->I, identifier) : nullptr;
    if (!d || !in)
        return (int)nrd::Result::INVALID_ARGUMENT;
    d->frameCounter = in->frame_counter;
    d->framesSinceReset = in->frames_since_reset;
    d->historyValid = in->history_valid != 0;
    return 0;
}

NRDHIP_API uint32_t orc_sizeof(uint32_t which) {
    switch (which) {
        case 0: return sizeof(nrd::CommonSettings);
        case 1: return sizeof(nrd::ReblurSettings);
        case 2: return sizeof(nrd::RelaxSettings);
        case 3: return sizeof(nrd::SigmaSettings);
        case 4: return sizeof(nrd::ReferenceSettings);
        case 5: return sizeof(nrdhip_create_desc);
        case 6: return sizeof(nrdhip_plane_info);
        case 7: return sizeof(nrdhip_dispatch_info);
        case 8: return sizeof(nrdhip_confidence_blur_desc);
        case 9: return sizeof(nrdhip_unpack_desc);
        case 10: return sizeof(nrdhip_taa_desc);
    }
    return 0;
}

NRDHIP_API const char* orc_last_error(nrdhip_instance* inst) { return inst ? inst->I.error.c_str() : g_createError.c_str(); }

// scalar helper exports for unit tests of the encodings (tests/test_oracle_math.py)
NRDHIP_API uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
NRDHIP_API float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
NRDHIP_API float orc_exp2(float x) { return exp2_poly(x); }
NRDHIP_API float orc_log2(float x) { return log2_poly(x); }
NRDHIP_API float orc_atan(float x) { return atan_pos(x); }
// (arrays: the accuracy tests of the default flavour's per-tap sequences run over millions of arguments)
NRDHIP_API void orc_sqrt1_array(const float* x, float* out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++)
        out[i] = sqrt1_unscaled_(x[i]) * SQRT1_SCALE;
}
NRDHIP_API void orc_exp2_neg_array(const float* x, float* out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++)
        out[i] = exp2_poly_neg(x[i]);
}
NRDHIP_API void orc_cbrt_array(const float* x, float* out, uint32_t n) { // (ledger row 19: TAA's pow(x, 0.333333))
    for (uint32_t i = 0; i < n; i++)
        out[i] = cbrt_pos_(x[i]);
}
NRDHIP_API uint32_t orc_pack_nr(float nx, float ny, float nz, float roughness, uint32_t mat) { return pack_normal_roughness({nx, ny, nz}, roughness, mat); }
NRDHIP_API void orc_unpack_nr(uint32_t p, float* out5) {
    NormalRoughness r = unpack_normal_roughness(p);
    out5[0] = r.n.x;
    out5[1] = r.n.y;
    out5[2] = r.n.z;
    out5[3] = r.roughness;
    out5[4] = (float)r.materialID;
}
NRDHIP_API float orc_hitdist_norm(float viewZ, const float* hp4, float roughness) { return reblur_hitdist_norm(absf(viewZ), hp4, roughness); }
NRDHIP_API void orc_ycocg(const float* rgb, float* out3, int inverse) {
    f3 r = inverse ? ycocg_to_linear({rgb[0], rgb[1], rgb[2]}) : linear_to_ycocg({rgb[0], rgb[1], rgb[2]});
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}
}
