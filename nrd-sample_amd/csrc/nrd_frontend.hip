// nrd_frontend.hip - the producer and the consumer either side of the denoiser as gfx950 kernels built on the host+device helper
// API include/nrd_frontend.h (the HIP twin of upstream's NRD.hlsli):
//   nrdhip_frontend_pack = what Shaders/TraceOpaque.cs.hlsl does with a path-tracing result before it becomes denoiser input
//                          (:421 hit distance normalisation, :657 normal / roughness / materialID, :738-757 radiance texels in
//                          NORMAL / OCCLUSION / SH / DIRECTIONAL_OCCLUSION mode for REBLUR and RELAX, :800-801 SIGMA inputs):
//                          raw fp32 planes in, the NRD input planes in their formats out. A C++ host needs no Python-made planes.
//   nrdhip_compose       = the rest of Shaders/Composition.cs.hlsl after the NRD decode (:92-107 NRD_SG_ReJitter in SH mode,
//                          :183-188 material re-modulation with NRD_MaterialFactors, hair excepted: RaytracingShared.hlsli:925-936)
// Both are streaming, one pixel per lane, HBM-bound; no LDS, no MFMA.
#include "nrd_device.h"

#include <cstring>

#include "../../include/nrd_frontend.h"
#include "../../include/nrdhip.h"

namespace nrdhip {
namespace {

using namespace nrd_fe;

struct PackParams {
    int W, H, mode, relax, sanitize;
    float hp[4], tanSun;
    PlaneRef normal, materialID, viewz, diff, spec, diffDir, specDir, shadow;
    PlaneRef outNR, outDiff, outSpec, outDiff1, outSpec1, outPen, outTransl;
};

NRD_DEV float4_ ldf4(const PlaneRef& P, int x, int y) {
    float4 v = ld<float4>(P, x, y, 16);
    return {v.x, v.y, v.z, v.w};
}
NRD_DEV void st_h4(const PlaneRef& P, int x, int y, float4_ v) {
    half4_ h = NRD_PackHalf4(v);
    st<uint2>(P, x, y, 8, uint2{h.lo, h.hi});
}

__global__ __launch_bounds__(256) void k_frontend_pack(const PackParams p) {
    int x = (int)(blockIdx.x * 64 + threadIdx.x), y = (int)(blockIdx.y * 4 + threadIdx.y);
    if (x >= p.W || y >= p.H)
        return;
    float roughness = 1.0f;
    if (p.normal.p) {
        float4_ n = ldf4(p.normal, x, y); // {world normal, roughness}
        roughness = n.w;
        if (p.outNR.p)
            st<uint32_t>(p.outNR, x, y, 4, NRD_FrontEnd_PackNormalAndRoughness({n.x, n.y, n.z}, n.w, p.materialID.p ? ld<float>(p.materialID, x, y, 4) : 0.0f));
    }
    const float viewZ = p.viewz.p ? ld<float>(p.viewz, x, y, 4) : 0.0f;
    const bool san = p.sanitize != 0;
#pragma unroll
    for (int sig = 0; sig < 2; sig++) {
        const PlaneRef& in = sig ? p.spec : p.diff;
        const PlaneRef& out = sig ? p.outSpec : p.outDiff;
        if (!in.p || !out.p)
            continue;
        float4_ r = ldf4(in, x, y); // {radiance, hit distance in world units}
        const PlaneRef& dirP = sig ? p.specDir : p.diffDir;
        const PlaneRef& out1 = sig ? p.outSpec1 : p.outDiff1;
        float3_ dir = {0.0f, 0.0f, 1.0f};
        if (dirP.p) {
            float4_ d = ldf4(dirP, x, y);
            dir = {d.x, d.y, d.z};
        }
        if (p.relax) {
            if (p.mode == NRDHIP_UNPACK_SH) {
                float4_ sh1;
                st_h4(out, x, y, RELAX_FrontEnd_PackSh({r.x, r.y, r.z}, r.w, dir, sh1, san));
                if (out1.p)
                    st_h4(out1, x, y, sh1);
            } else {
                st_h4(out, x, y, RELAX_FrontEnd_PackRadianceAndHitDist({r.x, r.y, r.z}, r.w, san));
            }
            continue;
        }
        float normHitDist = REBLUR_FrontEnd_GetNormHitDist(r.w, viewZ, p.hp, sig ? roughness : 1.0f);
        if (p.mode == NRDHIP_UNPACK_OCCLUSION) { // the normalised hit distance alone, R16_UNORM
            st<uint16_t>(out, x, y, 2, (uint16_t)__builtin_floorf(fma_(sat(normHitDist), 65535.0f, 0.5f)));
        } else if (p.mode == NRDHIP_UNPACK_SH) {
            float4_ sh1;
            st_h4(out, x, y, REBLUR_FrontEnd_PackSh({r.x, r.y, r.z}, normHitDist, dir, sh1, san));
            if (out1.p)
                st_h4(out1, x, y, sh1);
        } else if (p.mode == NRDHIP_PACK_DIRECTIONAL_OCCLUSION) {
            st_h4(out, x, y, REBLUR_FrontEnd_PackDirectionalOcclusion(dir, normHitDist, san));
        } else {
            st_h4(out, x, y, REBLUR_FrontEnd_PackRadianceAndNormHitDist({r.x, r.y, r.z}, normHitDist, san));
        }
    }
    if (p.shadow.p) {
        float4_ s = ldf4(p.shadow, x, y); // {distance to occluder (>= NRD_FP16_MAX: miss), translucency rgb}
        if (p.outPen.p)
            st<uint16_t>(p.outPen, x, y, 2, NRD_FloatToHalf(SIGMA_FrontEnd_PackPenumbra(s.x, p.tanSun)));
        if (p.outTransl.p)
            st<uint32_t>(p.outTransl, x, y, 4, NRD_PackUnorm8x4(SIGMA_FrontEnd_PackTranslucency(s.x, {s.y, s.z, s.w})));
    }
}

struct ComposeParams {
    int W, H, sh, relax;
    uint32_t hairMat;
    PlaneRef diff, spec, dSh0, dSh1, sSh0, sSh1, nr, viewz, bcm, outDiff, outSpec;
    float v2w[9], frustum[4], invW, invH;
    // base colour / metalness arrive as 8-bit codes: the 256 values NRD_SrgbToLinear( code / 255 ) and code / 255 can take, computed once on the
    // host with the very functions the kernel used to call per pixel (three divisions + three pow through log2 / exp2 polynomials: ~165 of
    // the pass's instructions) and read here from the argument segment
    float srgbLut[256], unorm8Lut[256];
};

NRD_DEV float4_ ld_h4(const PlaneRef& P, int x, int y) {
    uint2 t = ld<uint2>(P, x, y, 8);
    return NRD_UnpackHalf4({t.x, t.y});
}

__global__ __launch_bounds__(256) void k_compose(const ComposeParams p) {
    int x = (int)(blockIdx.x * 64 + threadIdx.x), y = (int)(blockIdx.y * 4 + threadIdx.y);
    if (x >= p.W || y >= p.H)
        return;
    float materialID;
    float4_ nr = NRD_FrontEnd_UnpackNormalAndRoughness(ld<uint32_t>(p.nr, x, y, 4), materialID);
    const float3_ N = {nr.x, nr.y, nr.z};
    const float roughness = nr.w;
    float u = ((float)x + 0.5f) * p.invW, v = ((float)y + 0.5f) * p.invH;
    // V = normalize( RotateVector( gViewToWorld, -Xv ) ) (Composition.cs.hlsl:74-75); the direction does not depend on viewZ
    float3_ Xv = {fe_fma(u, p.frustum[2], p.frustum[0]), fe_fma(v, p.frustum[3], p.frustum[1]), 1.0f};
    float3_ Xn = {-Xv.x, -Xv.y, -Xv.z};
    float3_ V = fe_normalize({fe_fma(p.v2w[2], Xn.z, fe_fma(p.v2w[1], Xn.y, p.v2w[0] * Xn.x)), fe_fma(p.v2w[5], Xn.z, fe_fma(p.v2w[4], Xn.y, p.v2w[3] * Xn.x)),
                              fe_fma(p.v2w[8], Xn.z, fe_fma(p.v2w[7], Xn.y, p.v2w[6] * Xn.x))});
    float4_ diff = p.diff.p ? ld_h4(p.diff, x, y) : float4_{0, 0, 0, 0}, spec = p.spec.p ? ld_h4(p.spec, x, y) : float4_{0, 0, 0, 0};
    if (p.sh && p.dSh0.p && p.sSh0.p) { // regain micro-details & jittering (:92-107)
        float4_ d0 = ld_h4(p.dSh0, x, y), d1 = ld_h4(p.dSh1, x, y), s0 = ld_h4(p.sSh0, x, y), s1 = ld_h4(p.sSh1, x, y);
        NRD_SG dSg = p.relax ? RELAX_BackEnd_UnpackSh(d0, {d1.x, d1.y, d1.z}) : REBLUR_BackEnd_UnpackSh(d0, {d1.x, d1.y, d1.z});
        NRD_SG sSg = p.relax ? RELAX_BackEnd_UnpackSh(s0, {s1.x, s1.y, s1.z}) : REBLUR_BackEnd_UnpackSh(s0, {s1.x, s1.y, s1.z});
        const int ox[4] = {1, -1, 0, 0}, oy[4] = {0, 0, 1, -1}; // e, w, n, s
        float Zn[4];
        float3_ Nn[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int px = imin(imax(x + ox[i], 0), p.W - 1), py = imin(imax(y + oy[i], 0), p.H - 1);
            float4_ t = NRD_FrontEnd_UnpackNormalAndRoughness(ld<uint32_t>(p.nr, px, py, 4));
            Nn[i] = {t.x, t.y, t.z};
            Zn[i] = ld<float>(p.viewz, px, py, 4);
        }
        float dScale, sScale;
        NRD_SG_ReJitter(dSg, sSg, V, roughness, ld<float>(p.viewz, x, y, 4), Zn, N, Nn, dScale, sScale);
        diff = {diff.x * dScale, diff.y * dScale, diff.z * dScale, diff.w};
        spec = {spec.x * sScale, spec.y * sScale, spec.z * sScale, spec.w};
    }
    // material modulation (:183-188): convert radiance back into irradiance
    float3_ diffFactor = {1.0f, 1.0f, 1.0f}, specFactor = {1.0f, 1.0f, 1.0f};
    if (p.bcm.p && (uint32_t)materialID != p.hairMat) {
        uint32_t b = ld<uint32_t>(p.bcm, x, y, 4);
        float3_ baseColor = {p.srgbLut[b & 255u], p.srgbLut[(b >> 8) & 255u], p.srgbLut[(b >> 16) & 255u]};
        float3_ albedo, Rf0;
        NRD_ConvertBaseColorMetalnessToAlbedoRf0(baseColor, p.unorm8Lut[b >> 24], albedo, Rf0);
        NRD_MaterialFactors(N, V, albedo, Rf0, roughness, diffFactor, specFactor);
    }
    if (p.outDiff.p)
        st_h4(p.outDiff, x, y, {diff.x * diffFactor.x, diff.y * diffFactor.y, diff.z * diffFactor.z, diff.w});
    if (p.outSpec.p)
        st_h4(p.outSpec, x, y, {spec.x * specFactor.x, spec.y * specFactor.y, spec.z * specFactor.z, spec.w});
}

PlaneRef plane(const void* p, uint32_t pitch, uint16_t w, uint16_t h) { return PlaneRef{(uint8_t*)p, pitch, w, h}; }

} // namespace
} // namespace nrdhip

extern "C" {

NRDHIP_API int nrdhip_frontend_pack(const nrdhip_frontend_pack_desc* d, void* hip_stream) {
    using namespace nrdhip;
    if (!d || !d->width || !d->height || d->mode > NRDHIP_PACK_DIRECTIONAL_OCCLUSION)
        return 2;
    if ((d->out_normal_roughness && !d->normal) || ((d->out_diff || d->out_spec) && !d->relax && !d->viewz) || ((d->out_penumbra || d->out_translucency) && !d->shadow))
        return 2;
    PackParams p = {};
    p.W = d->width;
    p.H = d->height;
    p.mode = (int)d->mode;
    p.relax = d->relax ? 1 : 0;
    p.sanitize = d->sanitize ? 1 : 0;
    for (int i = 0; i < 4; i++)
        p.hp[i] = d->hit_distance_parameters[i];
    p.tanSun = d->tan_of_light_angular_radius;
    const uint16_t w = d->width, h = d->height;
    p.normal = plane(d->normal, d->normal_pitch, w, h);
    p.materialID = plane(d->material_id, d->material_id_pitch, w, h);
    p.viewz = plane(d->viewz, d->viewz_pitch, w, h);
    p.diff = plane(d->diff, d->diff_pitch, w, h);
    p.spec = plane(d->spec, d->spec_pitch, w, h);
    p.diffDir = plane(d->diff_direction, d->diff_direction_pitch, w, h);
    p.specDir = plane(d->spec_direction, d->spec_direction_pitch, w, h);
    p.shadow = plane(d->shadow, d->shadow_pitch, w, h);
    p.outNR = plane(d->out_normal_roughness, d->out_normal_roughness_pitch, w, h);
    p.outDiff = plane(d->out_diff, d->out_diff_pitch, w, h);
    p.outSpec = plane(d->out_spec, d->out_spec_pitch, w, h);
    p.outDiff1 = plane(d->out_diff_sh1, d->out_diff_sh1_pitch, w, h);
    p.outSpec1 = plane(d->out_spec_sh1, d->out_spec_sh1_pitch, w, h);
    p.outPen = plane(d->out_penumbra, d->out_penumbra_pitch, w, h);
    p.outTransl = plane(d->out_translucency, d->out_translucency_pitch, w, h);
    dim3 grid((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), 1);
    hipLaunchKernelGGL(k_frontend_pack, grid, dim3(64, 4, 1), 0, (hipStream_t)hip_stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

NRDHIP_API int nrdhip_compose(const nrdhip_compose_desc* d, void* hip_stream) {
    using namespace nrdhip;
    if (!d || !d->width || !d->height || !d->normal_roughness || (!d->out_diff && !d->out_spec))
        return 2;
    if (d->sh && (!d->diff_sh0 || !d->diff_sh1 || !d->spec_sh0 || !d->spec_sh1 || !d->viewz))
        return 2;
    ComposeParams p = {};
    const uint16_t w = d->width, h = d->height;
    p.W = w;
    p.H = h;
    p.sh = d->sh ? 1 : 0;
    p.relax = d->relax ? 1 : 0;
    p.hairMat = d->hair_material_id;
    p.diff = plane(d->diff, d->diff_pitch, w, h);
    p.spec = plane(d->spec, d->spec_pitch, w, h);
    p.dSh0 = plane(d->diff_sh0, d->diff_sh0_pitch, w, h);
    p.dSh1 = plane(d->diff_sh1, d->diff_sh1_pitch, w, h);
    p.sSh0 = plane(d->spec_sh0, d->spec_sh0_pitch, w, h);
    p.sSh1 = plane(d->spec_sh1, d->spec_sh1_pitch, w, h);
    p.nr = plane(d->normal_roughness, d->normal_roughness_pitch, w, h);
    p.viewz = plane(d->viewz, d->viewz_pitch, w, h);
    p.bcm = plane(d->base_color_metalness, d->base_color_metalness_pitch, w, h);
    p.outDiff = plane(d->out_diff, d->out_diff_pitch, w, h);
    p.outSpec = plane(d->out_spec, d->out_spec_pitch, w, h);
    for (int i = 0; i < 9; i++)
        p.v2w[i] = d->view_to_world[i];
    for (int i = 0; i < 4; i++)
        p.frustum[i] = d->camera_frustum[i];
    p.invW = d->inv_rect_size[0];
    p.invH = d->inv_rect_size[1];
    static const struct Luts {
        float srgb[256], unorm8[256];
        Luts() {
            for (int i = 0; i < 256; i++) {
                unorm8[i] = (float)i / 255.0f;
                srgb[i] = NRD_SrgbToLinear((float)i / 255.0f);
            }
        }
    } luts;
    std::memcpy(p.srgbLut, luts.srgb, sizeof(p.srgbLut));
    std::memcpy(p.unorm8Lut, luts.unorm8, sizeof(p.unorm8Lut));
    dim3 grid((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), 1);
    hipLaunchKernelGGL(k_compose, grid, dim3(64, 4, 1), 0, (hipStream_t)hip_stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
}
