// tests/host/band_bounds_host.cpp - prints nrd::TiledIntegration::BandBounds / BandOf for the cases given on stdin, one per line:
//   frameHeight world minRows halo nCost cost[0] .. cost[nCost-1]          (nCost = 0: even split)
// output per case: "bounds b0 .. bworld | band(rank): frameH row0 ownFirst ownRows localH ..." (tests/test_band_layout.py compares it
// with nrd-sample_amd/tiler.py band_bounds / band_layout)
#include "../../include/NRDIntegration.h"

#include <cstdio>
#include <vector>

int main() {
    int frameH, world, minRows, halo, n;
    while (std::scanf("%d %d %d %d %d", &frameH, &world, &minRows, &halo, &n) == 5) {
        std::vector<float> cost((size_t)n);
        for (int i = 0; i < n; i++)
            if (std::scanf("%f", &cost[(size_t)i]) != 1)
                return 2;
        std::vector<int32_t> bounds((size_t)world + 1);
        if (!nrd::TiledIntegration::BandBounds((uint16_t)frameH, world, bounds.data(), n ? cost.data() : nullptr, (uint32_t)minRows)) {
            std::printf("refused\n");
            continue;
        }
        std::printf("bounds");
        for (int32_t b : bounds)
            std::printf(" %d", b);
        std::printf(" |");
        for (int r = 0; r < world; r++) {
            int32_t band[4];
            uint16_t localH = 0;
            nrd::TiledIntegration::BandOf((uint16_t)frameH, world, r, (uint32_t)halo, band, localH, bounds.data());
            std::printf(" %d %d %d %d %d", band[0], band[1], band[2], band[3], (int)localH);
        }
        std::printf("\n");
    }
    return 0;
}
