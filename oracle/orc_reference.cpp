// orc_reference.cpp - CPU ORACLE of nrd::Denoiser::REFERENCE (test infrastructure; parity unpinned).
//
// Contract (reference call sites): Source/NRDSample.cpp:484-485 binds the SAME texture to IN_SIGNAL and
// OUT_SIGNAL (in place), :1665-1667 feeds ReferenceSettings.maxAccumulatedFrameNum, :4213-4224 records it
// after SetCommonSettings(splitScreen), :3864 CLEAR_AND_RESTART resets the history.
// Algorithm: hist = lerp(hist, in, 1 / (1 + min(frames, maxAccumulatedFrameNum))) in an
// RGBA32F history; out = hist; columns left of splitScreen show the input.
#include "orc_core.h"

namespace orc {

void reference_describe(DenoiserState&, std::vector<PoolPlane>& perm, std::vector<PoolPlane>&) {
    perm.push_back({"REFERENCE::History", (uint32_t)nrd::Format::RGBA32_SFLOAT, 16, 1});
}

void reference_build(Instance&, DenoiserState& d) {
    Pass p;
    p.name = "REFERENCE::TemporalAccumulation";
    p.kernel = "nrd_reference_accumulate";
    p.haloRows = 0;
    p.bytesPerPixel = 8 + 16 + 16 + 8;
    p.read = {enc_slot(nrd::ResourceType::IN_SIGNAL), enc_perm(d.permBase)};
    p.written = {enc_perm(d.permBase), enc_slot(nrd::ResourceType::OUT_SIGNAL)};
    p.run = [](Instance& I, DenoiserState& d, const Consts& c, int y0, int y1) {
        const Plane& in = I.slots[(size_t)nrd::ResourceType::IN_SIGNAL];
        const Plane& out = I.slots[(size_t)nrd::ResourceType::OUT_SIGNAL];
        const Plane& hist = I.perm[d.permBase];
        bool restart = c.reset || !d.historyValid;
        uint32_t n = std::min(d.framesSinceReset, d.reference.maxAccumulatedFrameNum);
        float w = restart ? 1.0f : 1.0f / (1.0f + (float)n);
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < c.W; x++) {
                f4 s = ld_h4(in, x, y);
                f4 h;
                if (restart)
                    h = s;
                else {
                    h = {ld_f32(hist, x, y, 0), ld_f32(hist, x, y, 4), ld_f32(hist, x, y, 8), ld_f32(hist, x, y, 12)};
                    h = {lerpf(h.x, s.x, w), lerpf(h.y, s.y, w), lerpf(h.z, s.z, w), lerpf(h.w, s.w, w)};
                }
                st_f32(hist, x, y, h.x, 0);
                st_f32(hist, x, y, h.y, 4);
                st_f32(hist, x, y, h.z, 8);
                st_f32(hist, x, y, h.w, 12);
                float u = ((float)x + 0.5f) * c.invW;
                st_h4(out, x, y, u < c.splitScreen ? s : h);
            }
    };
    d.passes.push_back(p);
}

} // namespace orc
