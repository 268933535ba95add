"""The tile -> XCD traversal of the kernels (nrd_device.h: xcd_tile), exercised through the host-emulated build of the UNMODIFIED
kernel sources: every tile of a launch is visited exactly once for any grid shape, workgroup b stays on XCD b mod 8's share, every
XCD visits every column band and every block row (the load-balance property the rotation exists for), and consecutive workgroups
of an XCD stay inside one block (the locality property)."""
import ctypes

import pytest


@pytest.fixture(scope="module")
def emu_lib(emulated):
    lib = emulated.lib
    lib.nrdhip_debug_tile_of.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.nrdhip_debug_tile_of.restype = ctypes.c_int
    lib.nrdhip_debug_tile_of_table.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                               ctypes.POINTER(ctypes.c_int)]
    lib.nrdhip_debug_tile_of_table.restype = ctypes.c_int
    lib.nrdhip_debug_grid_blocks.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.nrdhip_debug_grid_blocks.restype = ctypes.c_uint
    return lib


def walk(lib, tiles_x, tiles_y, tile_y0=0, reverse=0):
    n = lib.nrdhip_debug_grid_blocks(tiles_x, tiles_y)
    tx, ty = ctypes.c_int(), ctypes.c_int()
    out = []
    for b in range(n):
        if lib.nrdhip_debug_tile_of(tiles_x, tiles_y, tile_y0, b, reverse, ctypes.byref(tx), ctypes.byref(ty)):
            out.append((b, tx.value, ty.value))
    return n, out


@pytest.mark.parametrize("tiles", [(1, 1), (3, 2), (4, 3), (7, 9), (8, 8), (9, 17), (15, 8), (16, 1), (30, 34), (61, 5), (120, 68), (240, 135)])
@pytest.mark.parametrize("reverse", [0, 1])
def test_every_tile_exactly_once(emu_lib, tiles, reverse):
    tiles_x, tiles_y = tiles
    n, seen = walk(emu_lib, tiles_x, tiles_y, tile_y0=3, reverse=reverse)
    assert n % 8 == 0 and n >= tiles_x * tiles_y
    assert sorted((x, y) for _, x, y in seen) == [(x, y) for x in range(tiles_x) for y in range(3, 3 + tiles_y)]
    assert n <= 8 * ((tiles_x + 7) // 8) * (tiles_y + 8)  # spare workgroups: band rounding + at most one block row of slack per block row


def test_4k_balance_and_locality(emu_lib):
    tiles_x, tiles_y = 240, 135  # 3840 x 2160
    n, seen = walk(emu_lib, tiles_x, tiles_y)
    cols, rows = (tiles_x + 7) // 8, (tiles_y + 7) // 8
    per_xcd = {k: [] for k in range(8)}
    for b, x, y in seen:
        per_xcd[b % 8].append((b, x, y))
    for k, lst in per_xcd.items():
        bands = {x // cols for _, x, _ in lst}
        block_rows = {y // rows for _, _, y in lst}
        assert bands == set(range(8)) and block_rows == set(range((tiles_y + rows - 1) // rows)), k  # every band, every block row
        # one block per (XCD, block row): the band is a function of the block row, distinct XCDs take distinct bands there
        for by in block_rows:
            assert len({x // cols for _, x, y in lst if y // rows == by}) == 1
        # the XCD's workgroups in launch order walk top to bottom block by block, row-major inside
        order = [(y // rows, y, x) for _, x, y in sorted(lst)]
        assert order == sorted(order)
        share = len(lst) / (tiles_x * tiles_y)
        assert abs(share - 1 / 8) < 0.002
    for by in range((tiles_y + rows - 1) // rows):
        taken = sorted({(b % 8, x // cols) for b, x, y in seen if y // rows == by})
        assert len({band for _, band in taken}) == 8 and len({k for k, _ in taken}) == 8


def test_8k_strips(emu_lib):
    """bands wider than the strip target are walked in two column strips (7680 x 4320: bands of 60 tiles -> 2 x 30)"""
    tiles_x, tiles_y = 480, 270
    _, seen = walk(emu_lib, tiles_x, tiles_y)
    first = [(x, y) for b, x, y in seen if b % 8 == 0][:30 * 34]
    xs = {x for x, _ in first}
    assert max(xs) - min(xs) == 29  # the first strip of XCD 0's first block: 30 tiles wide, all 34 rows of the block before the next strip


def test_reverse_walks_each_xcd_back_to_front(emu_lib):
    """FrameConsts::reverse (HistoryFix): XCD k visits the same tiles as in a forward launch, in the opposite order - the first
    workgroups of the launch work on the LAST tile rows, which is what the writer before it left in the Infinity Cache"""
    tiles_x, tiles_y = 240, 135
    _, fwd = walk(emu_lib, tiles_x, tiles_y)
    _, rev = walk(emu_lib, tiles_x, tiles_y, reverse=1)
    for k in range(8):
        f = [(x, y) for b, x, y in fwd if b % 8 == k]
        r = [(x, y) for b, x, y in rev if b % 8 == k]
        assert r == f[::-1]
    assert min(y for b, x, y in rev if b < 8 * 240) >= tiles_y - 17 - 8  # the first 240 workgroups of every XCD: the bottom block row


@pytest.mark.parametrize("tiles", [(1, 1), (7, 9), (9, 17), (30, 34), (61, 5), (240, 135)])
def test_tile_table_lookup_equals_the_computed_traversal(emu_lib, tiles):
    """FrameConsts::tileTable (one scalar load per wave instead of xcd_tile_kj's five integer divisions): a table holding the FORWARD order
    with tile rows relative to tileY0 - what nrdhip.cpp tile_table uploads - must hand every workgroup the tile the computation hands it,
    in both directions and for any first tile row, spare workgroups included"""
    tiles_x, tiles_y = tiles
    n = emu_lib.nrdhip_debug_grid_blocks(tiles_x, tiles_y)
    tx, ty = ctypes.c_int(), ctypes.c_int()
    table = (ctypes.c_uint32 * n)()
    for b in range(n):
        ok = emu_lib.nrdhip_debug_tile_of(tiles_x, tiles_y, 0, b, 0, ctypes.byref(tx), ctypes.byref(ty))
        table[b] = (tx.value | (ty.value << 16)) if ok else 0xFFFFFFFF
    for tile_y0, reverse in ((0, 0), (5, 0), (0, 1), (11, 1)):
        for b in range(n):
            want = (emu_lib.nrdhip_debug_tile_of(tiles_x, tiles_y, tile_y0, b, reverse, ctypes.byref(tx), ctypes.byref(ty)), tx.value, ty.value)
            got = (emu_lib.nrdhip_debug_tile_of_table(table, tiles_x, tiles_y, tile_y0, b, reverse, ctypes.byref(tx), ctypes.byref(ty)), tx.value, ty.value)
            assert got[0] == want[0] and (not want[0] or got[1:] == want[1:]), (b, tile_y0, reverse, want, got)
