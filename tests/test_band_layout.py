"""Row bands of the tiler: whole tile rows, every rank its share (not a remainder), optionally balanced by a
per-tile-row cost profile; the Python layout (nrd-sample_amd/tiler.py) and the C++ one (nrd::TiledIntegration::BandBounds / BandOf,
include/NRDIntegration.h, compiled for the host here) must agree to the row."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def band_tool(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("band") / "band_bounds_host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "band_bounds_host.cpp"), "-o", out],
                   check=True)
    return out


def test_even_split_world8_4320(pkg):
    from nrd_sample_amd import tiler

    b = tiler.band_bounds(4320, 8)
    rows = [y - x for x, y in zip(b, b[1:])]
    assert rows == [544, 544, 544, 544, 544, 544, 528, 528]  # 270 tile rows = 6 x 34 + 2 x 33 (was 7 x 528 + 624)
    assert b[0] == 0 and b[-1] == 4320 and all(x % 16 == 0 for x in b)
    assert max(rows) / (4320 / 8) < 1.01
    for frame_h, world in ((2160, 2), (2160, 3), (1080, 4), (100, 2), (4320, 7)):
        bb = tiler.band_bounds(frame_h, world)
        rr = [y - x for x, y in zip(bb, bb[1:])]
        assert sum(rr) == frame_h and all(x % 16 == 0 for x in bb[:-1])
        assert max(rr) - min(rr) <= 16 + (16 - frame_h % 16) % 16  # shares differ by at most one tile row (+ the ragged last tile)


def test_cost_balanced_split(pkg):
    from nrd_sample_amd import tiler

    n = 270
    sky_rows = 76  # the bench scene: the upper 28 % of the frame is sky
    cost = [tiler.SKY_TILE_COST * 480.0] * sky_rows + [480.0] * (n - sky_rows)
    b = tiler.band_bounds(4320, 8, cost, min_rows=80)
    per_band = [sum(cost[x // 16:(y + 15) // 16]) for x, y in zip(b, b[1:])]
    assert max(per_band) / (sum(cost) / 8) < 1.04  # within one tile row of the ideal share
    even = tiler.band_bounds(4320, 8)
    per_even = [sum(cost[x // 16:(y + 15) // 16]) for x, y in zip(even, even[1:])]
    assert max(per_even) / (sum(cost) / 8) > 1.25  # what the even split would do on this scene
    assert all(y - x >= 80 for x, y in zip(b, b[1:]))
    # all the cost in a few rows: the minimum band height still holds, and an impossible request is refused
    spike = [0.0] * n
    spike[100] = 1.0
    bs = tiler.band_bounds(4320, 8, spike, min_rows=80)
    assert all(y - x >= 80 for x, y in zip(bs, bs[1:])) and bs[-1] == 4320
    with pytest.raises(ValueError):
        tiler.band_bounds(256, 8, None, min_rows=80)


def test_cpp_layout_matches_python(pkg, band_tool):
    from nrd_sample_amd import tiler

    rng = np.random.default_rng(5)
    cases = []
    for frame_h, world, min_rows, halo in ((4320, 8, 80, 80), (4320, 8, 16, 80), (2160, 2, 80, 80), (2160, 3, 144, 144), (1080, 4, 16, 16), (100, 2, 16, 32), (4320, 7, 80, 96)):
        n = (frame_h + 15) // 16
        cases.append((frame_h, world, min_rows, halo, None))
        for _ in range(3):
            c = rng.random(n).astype(np.float32)
            c[: rng.integers(0, n // 2)] *= 0.15
            cases.append((frame_h, world, min_rows, halo, [float(v) for v in c]))
    text = ""
    for frame_h, world, min_rows, halo, cost in cases:
        text += "%d %d %d %d %d %s\n" % (frame_h, world, min_rows, halo, len(cost or []), " ".join(repr(v) for v in (cost or [])))
    out = subprocess.run([band_tool], input=text, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(cases)
    for line, (frame_h, world, min_rows, halo, cost) in zip(out, cases):
        head, tail = line.split("|")
        cpp_bounds = [int(v) for v in head.split()[1:]]
        bounds = tiler.band_bounds(frame_h, world, cost, min_rows)
        assert cpp_bounds == bounds, (frame_h, world, min_rows, cost is not None)
        vals = [int(v) for v in tail.split()]
        for r in range(world):
            L = tiler.band_layout(frame_h, world, r, halo, bounds)
            assert vals[5 * r:5 * r + 5] == [frame_h, L["row0"], L["own_first"], L["own_rows"], L["local_h"]]


def test_band_of_refuses_what_cannot_be_cut(tmp_path):
    """nrd::TiledIntegration::BandOf reports failure instead of building a band from uninitialised bounds - fewer tile rows
    than ranks, bands shorter than the halo, more than 64 ranks, caller bounds that are not monotone multiples of 16 ending at the frame
    height; Recreate turns that into Result::INVALID_ARGUMENT before anything is allocated"""
    src = tmp_path / "band_of.cpp"
    src.write_text(r'''
#include "NRDIntegration.h"
#include <cstdio>
int main() {
    using T = nrd::TiledIntegration;
    int32_t band[4]; uint16_t lh = 0; int bad = 0;
    bad += T::BandOf(100, 8, 0, 16, band, lh) ? 1 : 0;              // 7 tile rows over 8 ranks
    bad += T::BandOf(1088, 8, 0, 144, band, lh) ? 2 : 0;            // 68 tile rows, 8 bands of >= 9 tile rows each do not fit
    bad += T::BandOf(4320, 65, 0, 16, band, lh) ? 4 : 0;            // more ranks than the bounds array holds
    bad += T::BandOf(4320, 8, 8, 16, band, lh) ? 8 : 0;             // rank out of range
    const int32_t notMonotone[3] = {0, 1088, 1000}, notAligned[3] = {0, 500, 1088}, wrongEnd[3] = {0, 544, 1080}, good[3] = {0, 544, 1088};
    bad += T::BandOf(1088, 2, 0, 80, band, lh, notMonotone) ? 16 : 0;
    bad += T::BandOf(1088, 2, 0, 80, band, lh, notAligned) ? 32 : 0;
    bad += T::BandOf(1088, 2, 0, 80, band, lh, wrongEnd) ? 64 : 0;
    bad += T::BandOf(1088, 2, 1, 80, band, lh, good) ? 0 : 128;     // ... and accepts a proper cut
    bad += (band[0] == 1088 && band[1] == 464 && band[2] == 80 && band[3] == 544 && lh == 624) ? 0 : 256;
    bad += T::BandOf(4320, 8, 7, 80, band, lh) ? 0 : 512;           // the even split of config 5
    std::printf("%d\n", bad);
    return bad;
}
''')
    exe = str(tmp_path / "band_of")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
