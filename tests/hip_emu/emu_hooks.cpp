// emu_hooks.cpp - TEST INFRASTRUCTURE: entry points that only the host-emulated build of the kernel sources has (tests/hip_emu). They sit
// here, not in the product sources: the tile a workgroup of the XCD traversal works on (tests/test_tile_traversal.py) and the gather trace
// of the emulation (tools/gather_locality.py). Compiled into tests/_emu/libnrdhip_emu*.so next to the product's translation units.
#include "../../nrd-sample_amd/csrc/nrd_device.h"

using namespace nrdhip;

// the tile that workgroup `block` of a launch over tilesX x tilesY tiles works on (returns 0 when the workgroup is a spare one), and the
// launch size
extern "C" __attribute__((visibility("default"))) int nrdhip_debug_tile_of(int tilesX, int tilesY, int tileY0, unsigned block, int reverse, int* tx, int* ty) {
    FrameConsts c = {};
    c.reverse = reverse;
    c.tilesX = tilesX;
    c.tilesY = tilesY;
    c.tileY0 = tileY0;
    hipemu::t_blockIdx = {block, 0u, 0u};
    return xcd_tile(c, *tx, *ty) ? 1 : 0;
}
// the same through a tile order table (FrameConsts::tileTable: `table` = entries j * 8 + k of the FORWARD launch with tile rows relative to
// tileY0, as nrdhip.cpp tile_table builds it) - the lookup path of xcd_tile: direction and tileY0 are applied on top of the table
extern "C" __attribute__((visibility("default"))) int nrdhip_debug_tile_of_table(const uint32_t* table, int tilesX, int tilesY, int tileY0, unsigned block, int reverse, int* tx,
                                                                              int* ty) {
    FrameConsts c = {};
    c.reverse = reverse;
    c.tilesX = tilesX;
    c.tilesY = tilesY;
    c.tileY0 = tileY0;
    c.tileTable = table;
    c.tilesPerXcd = xcd_grid_blocks(tilesX, tilesY) / 8;
    hipemu::t_blockIdx = {block, 0u, 0u};
    return xcd_tile(c, *tx, *ty) ? 1 : 0;
}
extern "C" __attribute__((visibility("default"))) unsigned nrdhip_debug_grid_blocks(int tilesX, int tilesY) { return (unsigned)xcd_grid_blocks(tilesX, tilesY); }
// read and reset the gather trace of the emulation {wave-level gather instructions, distinct 128-byte lines they touched, lane loads},
// then switch it on / off
extern "C" __attribute__((visibility("default"))) void nrdhip_debug_gather_trace(int on, double* out3) {
    hipemu::GatherTrace& g = hipemu::g_gatherTrace;
    if (out3) {
        out3[0] = g.instr;
        out3[1] = g.lines;
        out3[2] = g.laneLoads;
    }
    g.instr = g.lines = g.laneLoads = 0;
    g.on = on != 0;
}
