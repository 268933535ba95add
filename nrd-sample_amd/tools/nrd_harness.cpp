// nrd_harness.cpp - headless C++ twin of the NRD part of the reference sample, written against include/NRD*.h exactly the
// way Source/NRDSample.cpp is written against the reference's headers:
//   Sample::Initialize .... denoiser descs + Integration::Recreate            (Source/NRDSample.cpp:869-990)
//   Sample::RenderFrame ... CommonSettings fill, NewFrame, SetCommonSettings  (:3835-3879)
//                           shadow denoising (:4068-4084), opaque denoising (:4086-4126), reference accumulation (:4213-4227)
//   Sample::Denoise ....... ResourceSnapshot::SetResource for every slot, Integration::Denoise (:440-531)
// Inputs are raw plane files (written by tests/test_cpp_harness.py or any producer); outputs are written back as raw files.
//
//   nrd_harness <dir> <width> <height> <frames>
#include "../../include/NRDIntegration.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
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

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: nrd_harness <dir> <width> <height> <frames>\n");
        return 2;
    }
    std::string dir = argv[1];
    uint16_t w = (uint16_t)atoi(argv[2]), h = (uint16_t)atoi(argv[3]);
    int frames = atoi(argv[4]);
    using F = nrd::Format;
    using RT = nrd::ResourceType;

    // ---- Sample::Initialize: REBLUR + SIGMA + REFERENCE in one instance ----
    const nrd::DenoiserDesc denoisersDescs[] = {
        {NRD_ID(REBLUR_DIFFUSE_SPECULAR), nrd::Denoiser::REBLUR_DIFFUSE_SPECULAR},
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
    if (!loadPlane(dir + "/mv.bin", Mv) || !loadPlane(dir + "/normal_roughness.bin", Normal_Roughness) || !loadPlane(dir + "/viewz.bin", ViewZ) ||
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
        commonSettings.isHistoryConfidenceAvailable = false;

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
    m_NRD.Destroy(); // before the device goes away (:744-748)
    return 0;
}
