"""Oracle arithmetic pinned against independent references (numpy) and the pack/unpack round trips of SURVEY.md 8c (3)."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib(oracle):
    L = oracle.lib
    L.orc_f32_to_f16.restype = ctypes.c_uint16
    L.orc_f32_to_f16.argtypes = [ctypes.c_float]
    L.orc_f16_to_f32.restype = ctypes.c_float
    L.orc_f16_to_f32.argtypes = [ctypes.c_uint16]
    for n in ("orc_exp2", "orc_log2", "orc_atan"):
        getattr(L, n).restype = ctypes.c_float
        getattr(L, n).argtypes = [ctypes.c_float]
    L.orc_pack_nr.restype = ctypes.c_uint32
    L.orc_pack_nr.argtypes = [ctypes.c_float] * 4 + [ctypes.c_uint32]
    L.orc_unpack_nr.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_float)]
    L.orc_hitdist_norm.restype = ctypes.c_float
    L.orc_hitdist_norm.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.c_float]
    L.orc_ycocg.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    return L


def test_f16_to_f32_exhaustive(lib):
    bits = np.arange(65536, dtype=np.uint16)
    ref = bits.view(np.float16).astype(np.float32)
    got = np.array([lib.orc_f16_to_f32(int(b)) for b in bits], dtype=np.float32)
    ok = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert ok.all()


def test_f32_to_f16_round_to_nearest_even(lib):
    rng = np.random.default_rng(1)
    # every half value, its neighbours' midpoints (ties), and random floats incl. denormal / overflow ranges
    halves = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float32)
    mids = (halves[:-1].astype(np.float64) + halves[1:].astype(np.float64)) * 0.5
    vals = np.concatenate([halves, mids.astype(np.float32), np.nextafter(mids.astype(np.float32), np.float32(0)),
                           np.nextafter(mids.astype(np.float32), np.float32(1e9)),
                           np.exp(rng.uniform(-20, 12, 20000)).astype(np.float32), np.array([65504, 65519.9, 65520, 1e9, 0, 1e-9], np.float32)])
    vals = np.concatenate([vals, -vals])
    with np.errstate(over="ignore"):
        ref = vals.astype(np.float16).view(np.uint16)
    got = np.array([lib.orc_f32_to_f16(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got, ref)


def test_polynomial_transcendentals(lib):
    xs = np.linspace(-30, 30, 4001, dtype=np.float32)
    e = np.array([lib.orc_exp2(float(x)) for x in xs])
    assert np.max(np.abs(e / np.exp2(xs.astype(np.float64)) - 1)) < 5e-7
    ps = np.exp(np.linspace(-20, 20, 4001)).astype(np.float32)
    l2 = np.array([lib.orc_log2(float(x)) for x in ps])
    assert np.max(np.abs(l2 - np.log2(ps.astype(np.float64)))) < 2e-6
    at = np.linspace(0, 50, 2001, dtype=np.float32)
    a = np.array([lib.orc_atan(float(x)) for x in at])
    assert np.max(np.abs(a - np.arctan(at.astype(np.float64)))) < 2e-5


def test_normal_roughness_material_roundtrip(lib):
    rng = np.random.default_rng(2)
    n = rng.standard_normal((2000, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    out = (ctypes.c_float * 5)()
    worst = 0.0
    for i in range(len(n)):
        r = float(rng.uniform())
        mat = int(rng.integers(0, 4))
        p = lib.orc_pack_nr(float(n[i, 0]), float(n[i, 1]), float(n[i, 2]), r, mat)
        lib.orc_unpack_nr(p, out)
        got = np.array(out[:3])
        worst = max(worst, float(np.arccos(np.clip(np.dot(got, n[i]), -1, 1))))
        assert abs(out[3] - r) <= 0.5 / 1023 + 1e-7
        assert int(out[4]) == mat  # materialID exact for 0..3 (Shaders/Shared.hlsli:94-97)
        assert abs(np.linalg.norm(got) - 1) < 1e-5
    assert worst < 5e-3  # 10-bit octahedral quantisation (step 2/1023 per axis, stretched near the octant edges)


def test_pack_matches_generator(lib, pkg):
    """the numpy generator (inputs) and the oracle agree on the R10G10B10A2 packing"""
    rng = np.random.default_rng(3)
    n = rng.standard_normal((500, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    r = rng.uniform(size=500)
    m = rng.integers(0, 4, 500)
    ref = pkg.synth.pack_normal_roughness(n, r, m)
    same = 0
    for i in range(500):
        same += int(lib.orc_pack_nr(float(np.float32(n[i, 0])), float(np.float32(n[i, 1])), float(np.float32(n[i, 2])), float(np.float32(r[i])), int(m[i])) == int(ref[i]))
    assert same >= 490  # float32 vs float64 rounding may flip a 10-bit code on exact ties only


def test_ycocg_roundtrip_and_hitdist_norm(lib, pkg):
    rng = np.random.default_rng(4)
    o = (ctypes.c_float * 3)()
    b = (ctypes.c_float * 3)()
    for _ in range(500):
        c = rng.uniform(0, 50, 3).astype(np.float32)
        lib.orc_ycocg((ctypes.c_float * 3)(*c), o, 0)
        lib.orc_ycocg(o, b, 1)
        assert np.allclose(np.array(b[:]), c, rtol=2e-6, atol=1e-5)
    hp = (ctypes.c_float * 4)(3.0, 0.1, 20.0, -25.0)
    for z, r in ((1.0, 1.0), (10.0, 0.05), (55.0, 0.3), (0.2, 0.0)):
        ref = float(pkg.synth.reblur_hitdist_norm(np.float64(z), np.float64(r)))
        assert abs(lib.orc_hitdist_norm(z, hp, r) / ref - 1) < 1e-5


def test_per_tap_sequences_of_the_default_flavour(oracle):
    """the two approximations a spatial tap of the default build pays for (csrc/nrd_device.h == oracle/orc_math.h): the one-step square
    root of the normal weight's chord over EVERY squared code distance that can occur (0 .. 3 x 1023^2): relative error <= 6.6e-4, exact
    at 0, monotone enough to be a distance (never negative); the degree-3 exp2 of the hit-distance weight on [-126, 0]: relative error
    <= 8.1e-5, and what lies below the clamp stays at 2^-126"""
    L = oracle.lib
    fp = ctypes.POINTER(ctypes.c_float)
    for fn in (L.orc_sqrt1_array, L.orc_exp2_neg_array):
        fn.restype = None
        fn.argtypes = [fp, fp, ctypes.c_uint32]
    x = np.arange(0, 3 * 1023 * 1023 + 1, dtype=np.float32)
    out = np.empty_like(x)
    L.orc_sqrt1_array(x.ctypes.data_as(fp), out.ctypes.data_as(fp), x.size)
    ref = np.sqrt(x.astype(np.float64))
    assert out[0] == 0.0 and (out >= 0).all()
    assert np.abs(out[1:].astype(np.float64) / ref[1:] - 1.0).max() <= 6.6e-4
    e = np.concatenate([np.linspace(-126.0, 0.0, 2000001), [-200.0, -1e9]]).astype(np.float32)
    got = np.empty_like(e)
    L.orc_exp2_neg_array(e.ctypes.data_as(fp), got.ctypes.data_as(fp), e.size)
    want = np.exp2(np.maximum(e.astype(np.float64), -126.0))
    assert np.abs(got.astype(np.float64) / want - 1.0).max() <= 8.1e-5


def test_cube_root_of_the_lab_conversion(oracle):
    """ledger row 19 (csrc/nrd_device.h == oracle/orc_math.h cbrt_pos_): TAA's XyzToLab takes pow(x, 0.333333) (Shaders/Taa.cs.hlsl:45-47)
    as the cube root - Newton on the inverse cube root, three steps. Over the range the conversion can see (x > 0.008856; XYZ of an fp16
    colour stays below 1e5): relative error <= 5e-7 against the cube root and <= 5e-6 against x ** 0.333333 in float64, monotone"""
    L = oracle.lib
    fp = ctypes.POINTER(ctypes.c_float)
    L.orc_cbrt_array.restype = None
    L.orc_cbrt_array.argtypes = [fp, fp, ctypes.c_uint32]
    x = np.sort(np.exp(np.random.default_rng(3).uniform(np.log(0.008856), np.log(1e6), 4000000)).astype(np.float32))
    got = np.empty_like(x)
    L.orc_cbrt_array(x.ctypes.data_as(fp), got.ctypes.data_as(fp), x.size)
    x64 = x.astype(np.float64)
    assert np.abs(got / np.cbrt(x64) - 1.0).max() <= 5e-7
    assert np.abs(got / x64 ** 0.333333 - 1.0).max() <= 5e-6
    assert (np.diff(got.astype(np.float64)) >= -1e-6 * got[1:]).all()  # monotone to within twice its error bound
    # exact cubes come back to within one ULP
    c = np.arange(1, 60, dtype=np.float32)
    out = np.empty_like(c)
    cubes = (c * c * c).astype(np.float32)
    L.orc_cbrt_array(cubes.ctypes.data_as(fp), out.ctypes.data_as(fp), c.size)
    assert np.abs(out / c - 1.0).max() <= 2.4e-7


def test_unorm10_decode_sequence_is_the_ieee_quotient():
    """nrd_device.h unorm10_: q = x * r; q' = fma(fma(-q, 1023, x), r, q) with r = fl(1/1023) equals the correctly rounded x / 1023
    (what the oracle's `/` computes) for every 10-bit x - checked in exact rational arithmetic with one rounding per operation"""
    import math
    from fractions import Fraction as F

    def rnd32(v):
        if v == 0:
            return F(0)
        sgn, v = (1 if v > 0 else -1), abs(v)
        e = math.floor(math.log2(v))
        while F(2) ** e > v:
            e -= 1
        while F(2) ** (e + 1) <= v:
            e += 1
        ulp = F(2) ** (max(e, -126) - 23)
        n = v / ulp
        fl = n.numerator // n.denominator
        rem = n - fl
        if rem > F(1, 2) or (rem == F(1, 2) and fl % 2 == 1):
            fl += 1
        return sgn * fl * ulp

    r = F(float(np.float32(1.0) / np.float32(1023.0)))
    for x in range(1024):
        q = rnd32(F(x) * r)
        q2 = rnd32(rnd32(-q * 1023 + x) * r + q)
        assert q2 == rnd32(F(x, 1023)), x
        assert float(q2) == float(np.float32(x) / np.float32(1023.0))
