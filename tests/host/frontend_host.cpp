// TEST INFRASTRUCTURE: include/nrd_frontend.h compiled by a plain host compiler (g++, no HIP) and exposed to the tests through ctypes -
// proves the header is host + device, gives the known-answer tests something to call, and provides the host-side reference the
// kernels nrdhip_frontend_pack / nrdhip_compose must match bit for bit (same functions, same order of operations).
#include "../../include/nrd_frontend.h"

#include <cstring>

using namespace nrd_fe;

extern "C" {
uint32_t fe_pack_nr(float x, float y, float z, float r, float m) { return NRD_FrontEnd_PackNormalAndRoughness({x, y, z}, r, m); }
void fe_unpack_nr(uint32_t p, float* out5) {
    float4_ v = NRD_FrontEnd_UnpackNormalAndRoughness(p, out5[4]);
    out5[0] = v.x, out5[1] = v.y, out5[2] = v.z, out5[3] = v.w;
}
uint16_t fe_f2h(float f) { return NRD_FloatToHalf(f); }
float fe_h2f(uint16_t h) { return NRD_HalfToFloat(h); }
float fe_norm_hit(float h, float z, const float* hp, float rough) { return REBLUR_FrontEnd_GetNormHitDist(h, z, hp, rough); }
float fe_spec_avg(const float* h, int n) {
    float acc = NRD_FrontEnd_SpecHitDistAveraging_Begin();
    for (int i = 0; i < n; i++)
        NRD_FrontEnd_SpecHitDistAveraging_Add(acc, h[i]);
    NRD_FrontEnd_SpecHitDistAveraging_End(acc);
    return acc;
}
float fe_penumbra(float d, float t) { return SIGMA_FrontEnd_PackPenumbra(d, t); }
uint32_t fe_translucency(float d, float r, float g, float b) { return NRD_PackUnorm8x4(SIGMA_FrontEnd_PackTranslucency(d, {r, g, b})); }
void fe_material_factors(const float* N, const float* V, const float* albedo, const float* Rf0, float rough, float* out6) {
    float3_ df, sf;
    NRD_MaterialFactors({N[0], N[1], N[2]}, {V[0], V[1], V[2]}, {albedo[0], albedo[1], albedo[2]}, {Rf0[0], Rf0[1], Rf0[2]}, rough, df, sf);
    out6[0] = df.x, out6[1] = df.y, out6[2] = df.z, out6[3] = sf.x, out6[4] = sf.y, out6[5] = sf.z;
}
float fe_exp2(float x) { return nrd_fe::fe_exp2(x); }
float fe_log2(float x) { return nrd_fe::fe_log2(x); }

// one frame through the same helper calls the kernel k_frontend_pack makes (NORMAL / SH / OCCLUSION / DIRECTIONAL_OCCLUSION, REBLUR / RELAX)
void fe_pack_frame(int n, int mode, int relax, const float* hp, float tanSun, const float* normal, const float* mat, const float* viewz, const float* diff, const float* spec,
                   const float* ddir, const float* sdir, const float* shadow, uint32_t* outNR, uint16_t* outDiff, uint16_t* outSpec, uint16_t* outDiff1, uint16_t* outSpec1,
                   uint16_t* outPen, uint32_t* outTransl) {
    auto put = [](uint16_t* dst, int i, float4_ v) {
        half4_ h = NRD_PackHalf4(v);
        std::memcpy(dst + 4 * i, &h, 8);
    };
    for (int i = 0; i < n; i++) {
        float rough = normal[4 * i + 3];
        outNR[i] = NRD_FrontEnd_PackNormalAndRoughness({normal[4 * i], normal[4 * i + 1], normal[4 * i + 2]}, rough, mat[i]);
        for (int sig = 0; sig < 2; sig++) {
            const float* in = sig ? spec : diff;
            const float* dirs = sig ? sdir : ddir;
            uint16_t* out = sig ? outSpec : outDiff;
            uint16_t* out1 = sig ? outSpec1 : outDiff1;
            float3_ rad = {in[4 * i], in[4 * i + 1], in[4 * i + 2]}, dir = {dirs[4 * i], dirs[4 * i + 1], dirs[4 * i + 2]};
            float hit = in[4 * i + 3];
            float4_ sh1 = {0, 0, 0, 0};
            if (relax) {
                if (mode == 2) {
                    put(out, i, RELAX_FrontEnd_PackSh(rad, hit, dir, sh1));
                    put(out1, i, sh1);
                } else
                    put(out, i, RELAX_FrontEnd_PackRadianceAndHitDist(rad, hit));
                continue;
            }
            float nh = REBLUR_FrontEnd_GetNormHitDist(hit, viewz[i], hp, sig ? rough : 1.0f);
            if (mode == 1)
                out[i] = (uint16_t)__builtin_floorf(__builtin_fmaf(fe_sat(nh), 65535.0f, 0.5f));
            else if (mode == 2) {
                put(out, i, REBLUR_FrontEnd_PackSh(rad, nh, dir, sh1));
                put(out1, i, sh1);
            } else if (mode == 3)
                put(out, i, REBLUR_FrontEnd_PackDirectionalOcclusion(dir, nh));
            else
                put(out, i, REBLUR_FrontEnd_PackRadianceAndNormHitDist(rad, nh));
        }
        outPen[i] = NRD_FloatToHalf(SIGMA_FrontEnd_PackPenumbra(shadow[4 * i], tanSun));
        outTransl[i] = NRD_PackUnorm8x4(SIGMA_FrontEnd_PackTranslucency(shadow[4 * i], {shadow[4 * i + 1], shadow[4 * i + 2], shadow[4 * i + 3]}));
    }
}
}
