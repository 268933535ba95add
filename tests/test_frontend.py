"""the shader-side helper API as a host+device header (include/nrd_frontend.h), the
producer kernel built on it (nrdhip_frontend_pack) and the rest of the Composition consumer (nrdhip_compose: NRD_SG_ReJitter +
NRD_MaterialFactors re-modulation). Checked three ways: known answers of the header compiled by plain g++ (SURVEY 8c (3)
round-trip / known-answer tests), kernel == host header bit for bit, kernel vs independent numpy float64 restatements (<= 1 fp16
ULP). `emulated` runs the kernel sources on the CPU every round; `hip` is the real thing."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BACKENDS = ["emulated", pytest.param("hip", marks=pytest.mark.gpu)]
HP = (3.0, 0.1, 20.0, -25.0)


@pytest.fixture(scope="session")
def fe():
    """include/nrd_frontend.h through g++ (no HIP anywhere)"""
    out = os.path.join(ROOT, "tests", "_emu", "libfrontend_host.so")
    src = os.path.join(ROOT, "tests", "host", "frontend_host.cpp")
    hdr = os.path.join(ROOT, "include", "nrd_frontend.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", src, "-o", out], check=True)
    lib = C.CDLL(out)
    f, u32, u16 = C.c_float, C.c_uint32, C.c_uint16
    lib.fe_pack_nr.restype, lib.fe_pack_nr.argtypes = u32, [f] * 5
    lib.fe_unpack_nr.argtypes = [u32, C.POINTER(f)]
    lib.fe_f2h.restype, lib.fe_f2h.argtypes = u16, [f]
    lib.fe_h2f.restype, lib.fe_h2f.argtypes = f, [u16]
    lib.fe_norm_hit.restype, lib.fe_norm_hit.argtypes = f, [f, f, C.POINTER(f), f]
    lib.fe_spec_avg.restype, lib.fe_spec_avg.argtypes = f, [C.POINTER(f), C.c_int]
    lib.fe_penumbra.restype, lib.fe_penumbra.argtypes = f, [f, f]
    lib.fe_translucency.restype, lib.fe_translucency.argtypes = u32, [f] * 4
    lib.fe_material_factors.argtypes = [C.POINTER(f)] * 4 + [f, C.POINTER(f)]
    lib.fe_exp2.restype, lib.fe_exp2.argtypes = f, [f]
    lib.fe_log2.restype, lib.fe_log2.argtypes = f, [f]
    return lib


def fa(*v):
    return (C.c_float * len(v))(*v)


def test_known_answers_of_the_host_compiled_header(fe):
    # normal / roughness / materialID: R10G10B10A2, octahedral, round to nearest
    assert fe.fe_pack_nr(0, 0, 1, 1.0, 3) == 512 | (512 << 10) | (1023 << 20) | (3 << 30)
    assert fe.fe_pack_nr(1, 0, 0, 0.0, 0) == 1023 | (512 << 10)
    assert fe.fe_pack_nr(0, 0, -1, 0.5, 1) == 1023 | (1023 << 10) | (512 << 20) | (1 << 30)
    out = fa(0, 0, 0, 0, 0)
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = rng.standard_normal(3)
        n /= np.linalg.norm(n)
        r, m = rng.random(), rng.integers(0, 4)
        fe.fe_unpack_nr(fe.fe_pack_nr(*[float(v) for v in n], float(r), float(m)), out)
        assert np.dot(n, out[0:3]) > 1 - 2e-5 and abs(out[3] - r) <= 0.5 / 1023 + 1e-6 and out[4] == m  # 10-bit oct: < 0.4 deg
    # fp16 conversions: every half, and float -> half against numpy's round-to-nearest-even
    allh = np.arange(65536, dtype=np.uint16)
    ref = allh.view(np.float16).astype(np.float32)
    got = np.array([fe.fe_h2f(int(h)) for h in allh], np.float32)
    ok = np.isnan(ref) == np.isnan(got)
    assert ok.all() and np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])
    vals = np.concatenate([rng.standard_normal(4000) * 10.0 ** rng.integers(-9, 5, 4000), ref[np.isfinite(ref)][::37], [0.0, -0.0, 65504.0, 65520.0, 1e9, -1e9, 6.1e-5, 5.96e-8, 2.9e-8, 3.1e-8]]).astype(np.float32)
    want = np.clip(vals, -65504, 65504).astype(np.float16).view(np.uint16)
    got = np.array([fe.fe_f2h(float(v)) for v in vals], np.uint16)
    assert np.array_equal(got, want)
    # REBLUR hit distance normalisation: (A + |z| B) lerp(1, C, 2^(D r^2))
    assert abs(fe.fe_norm_hit(2.0, 10.0, fa(*HP), 1.0) - 2.0 / (4.0 * (1 + 19 * 2.0 ** -25)) ) < 1e-6
    assert abs(fe.fe_norm_hit(2.0, -10.0, fa(*HP), 0.0) - 2.0 / 80.0) < 1e-7
    assert fe.fe_norm_hit(1e9, 1.0, fa(*HP), 1.0) == 1.0 and fe.fe_norm_hit(0.0, 1.0, fa(*HP), 0.3) == 0.0
    # specular hit distance averaging is a soft MINIMUM: one path is returned as is, a near + a far hit stay near the near one
    assert abs(fe.fe_spec_avg(fa(0.37), 1) - 0.37) < 2e-6
    both = fe.fe_spec_avg(fa(0.1, 0.9), 2)
    assert 0.0999 < both <= 0.1 and fe.fe_spec_avg(fa(), 0) == 0.0
    assert fe.fe_spec_avg(fa(0.5, 0.5, 0.5, 0.5), 4) < 0.5 and fe.fe_spec_avg(fa(0.5, 0.5, 0.5, 0.5), 4) > 0.5 - 2.0 / 17 - 1e-6
    # SIGMA inputs
    assert fe.fe_penumbra(65504.0, 0.005) == 65504.0 and abs(fe.fe_penumbra(10.0, 0.005) - 0.05) < 1e-8 and fe.fe_penumbra(1e9 * 0 + 60000.0, 1.0) == 32768.0
    assert fe.fe_translucency(65504.0, 0.2, 0.4, 1.0) == 255 | (51 << 8) | (102 << 16) | (255 << 24)
    assert fe.fe_translucency(3.0, 0.0, 0.0, 0.0) == 0
    # material factors: in (0, 1], specular factor grows toward grazing angles on a smooth dielectric, a mirror metal reflects everything
    out6 = fa(*[0] * 6)
    facing, grazing = [], []
    for nov, acc in ((1.0, facing), (0.05, grazing)):
        v = (float(np.sqrt(1 - nov * nov)), 0.0, nov)
        fe.fe_material_factors(fa(0, 0, 1), fa(*v), fa(0.5, 0.5, 0.5), fa(0.04, 0.04, 0.04), 0.05, out6)
        acc += list(out6)
    assert all(0.0 < x <= 1.0 for x in facing + grazing) and grazing[3] > 3 * facing[3] and 0.02 < facing[3] < 0.08
    assert abs(facing[0] - ((1 - (facing[3] - 0.01) / 0.99) * 0.5 * 0.99 + 0.01)) < 1e-6  # diffFactor = (1 - Fenv) albedo 0.99 + 0.01
    fe.fe_material_factors(fa(0, 0, 1), fa(0, 0, 1), fa(0, 0, 0), fa(1, 1, 1), 0.0, out6)
    assert out6[3] > 0.95 and abs(out6[0] - 0.01) < 1e-6
    # the fixed polynomials behind them
    for x in (-20.5, -1.0, 0.0, 0.3, 7.25):
        assert abs(fe.fe_exp2(x) / 2.0 ** x - 1) < 3e-7
    for x in (1e-3, 0.5, 1.0, 3.7, 6e4):
        assert abs(fe.fe_log2(x) - np.log2(x)) < 3e-6


def random_frame(n, seed):
    rng = np.random.default_rng(seed)
    nrm = rng.standard_normal((n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    normal = np.concatenate([nrm, rng.random((n, 1))], 1).astype(np.float32)
    mat = rng.integers(0, 4, n).astype(np.float32)
    viewz = (rng.random(n) * 50 + 0.5).astype(np.float32) * np.where(rng.random(n) < 0.5, 1, -1).astype(np.float32)
    rad = lambda: np.concatenate([np.exp(rng.standard_normal((n, 3)) * 1.5), np.exp(rng.standard_normal((n, 1))) * 3], 1).astype(np.float32)
    diff, spec = rad(), rad()
    diff[::97, 0] = np.inf  # sanitised away
    spec[::89, 3] = np.nan
    dirs = lambda: np.concatenate([(lambda d: d / np.linalg.norm(d, axis=1, keepdims=True))(rng.standard_normal((n, 3))), np.zeros((n, 1))], 1).astype(np.float32)
    shadow = np.concatenate([np.where(rng.random((n, 1)) < 0.3, 65504.0, rng.random((n, 1)) * 40), rng.random((n, 3))], 1).astype(np.float32)
    return dict(normal=normal, material_id=mat, viewz=viewz, diff=diff, spec=spec, diff_direction=dirs(), spec_direction=dirs(), shadow=shadow)


def run_pack(pkg, backend, fr, w, h, mode, relax):
    import torch

    from nrd_sample_amd import sample_passes as sp

    dev = backend.device

    def up(a, ch):
        a = np.ascontiguousarray(a.reshape(h, w * ch))
        return torch.from_numpy(a).to(dev) if backend.is_device else a

    def z(dtype, ch):
        return torch.zeros((h, w * ch), dtype=getattr(torch, dtype), device=dev) if backend.is_device else np.zeros((h, w * ch), getattr(np, dtype))

    ins = {k: up(v, 1 if v.ndim == 1 else 4) for k, v in fr.items()}
    occ = mode == 1
    outs = dict(out_normal_roughness=z("int32", 1), out_diff=z("int16", 1 if occ else 4), out_spec=z("int16", 1 if occ else 4), out_diff_sh1=z("int16", 4),
                out_spec_sh1=z("int16", 4), out_penumbra=z("int16", 1), out_translucency=z("int32", 1))
    sp.frontend_pack(backend, w, h, mode=mode, relax=relax, hit_distance_parameters=HP, tan_of_light_angular_radius=0.00465, **ins, **outs)
    if backend.is_device:
        torch.cuda.synchronize()
        outs = {k: v.cpu().numpy() for k, v in outs.items()}
    return outs


@pytest.mark.parametrize("which", BACKENDS)
@pytest.mark.parametrize("mode,relax", [(0, False), (0, True), (1, False), (2, False), (2, True), (3, False)])
def test_pack_kernel_matches_the_host_header_bit_for_bit(request, pkg, fe, which, mode, relax):
    backend = request.getfixturevalue(which)
    w, h = 96, 40
    n = w * h
    fr = random_frame(n, 11 + mode)
    got = run_pack(pkg, backend, fr, w, h, mode, relax)
    P = lambda a, t: np.ascontiguousarray(a).ctypes.data_as(C.POINTER(t))
    nr, tr = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    d, s, d1, s1, pen = (np.zeros(n * 4, np.uint16) for _ in range(5))
    fe.fe_pack_frame.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float] + [C.POINTER(C.c_float)] * 8 + [C.POINTER(C.c_uint32)] + [C.POINTER(C.c_uint16)] * 5 + [C.POINTER(C.c_uint32)]
    fe.fe_pack_frame(n, mode, int(relax), fa(*HP), 0.00465, *[P(fr[k], C.c_float) for k in ("normal", "material_id", "viewz", "diff", "spec", "diff_direction", "spec_direction", "shadow")],
                     P(nr, C.c_uint32), P(d, C.c_uint16), P(s, C.c_uint16), P(d1, C.c_uint16), P(s1, C.c_uint16), P(pen, C.c_uint16), P(tr, C.c_uint32))
    assert np.array_equal(got["out_normal_roughness"].view(np.uint32).reshape(-1), nr)
    assert np.array_equal(got["out_translucency"].view(np.uint32).reshape(-1), tr)
    assert np.array_equal(got["out_penumbra"].view(np.uint16).reshape(-1), pen[:n])
    ch = 1 if (mode == 1 and not relax) else 4
    assert np.array_equal(got["out_diff"].view(np.uint16).reshape(-1), d[:n * ch]) and np.array_equal(got["out_spec"].view(np.uint16).reshape(-1), s[:n * ch])
    if mode == 2:
        assert np.array_equal(got["out_diff_sh1"].view(np.uint16).reshape(-1), d1) and np.array_equal(got["out_spec_sh1"].view(np.uint16).reshape(-1), s1)


def ulp16(a, b):
    a = a.view(np.int16).astype(np.int32)
    b = b.view(np.int16).astype(np.int32)
    a = np.where(a < 0, -32768 - a, a)
    b = np.where(b < 0, -32768 - b, b)
    return np.abs(a - b)


@pytest.mark.parametrize("which", BACKENDS)
def test_pack_kernel_against_numpy_float64(request, pkg, which):
    """independent restatement (nrd-sample_amd/synth.py helpers + plain numpy, float64): codes may differ by one step only where the
    float64 value sits on a rounding boundary of the float32 one"""
    from nrd_sample_amd import synth

    backend = request.getfixturevalue(which)
    w, h = 128, 32
    n = w * h
    fr = random_frame(n, 5)
    fr["diff"][::97, 0] = 1.0
    fr["spec"][::89, 3] = 1.0
    got = run_pack(pkg, backend, fr, w, h, 0, False)
    want_nr = synth.pack_normal_roughness(fr["normal"][:, :3].astype(np.float64), fr["normal"][:, 3].astype(np.float64), fr["material_id"].astype(np.uint32))
    gnr = got["out_normal_roughness"].view(np.uint32).reshape(-1)
    for shift in (0, 10, 20):
        dlt = np.abs(((gnr >> shift) & 1023).astype(np.int64) - ((want_nr >> shift) & 1023).astype(np.int64))
        assert dlt.max() <= 1 and (dlt > 0).mean() < 0.005
    assert np.array_equal(gnr >> 30, want_nr >> 30)
    for key, rough in (("diff", np.ones(n)), ("spec", fr["normal"][:, 3].astype(np.float64))):
        x = fr[key].astype(np.float64)
        yc = synth.linear_to_ycocg(x[:, :3])
        nh = np.clip(x[:, 3] / synth.reblur_hitdist_norm(fr["viewz"].astype(np.float64), rough, HP), 0, 1)
        want = np.concatenate([yc, nh[:, None]], 1).astype(np.float16)
        g = got["out_" + key].view(np.float16).reshape(n, 4)
        assert ulp16(g, want).max() <= 1
    pen = got["out_penumbra"].view(np.float16).reshape(-1).astype(np.float64)
    miss = fr["shadow"][:, 0] >= 65504
    assert (pen[miss] == 65504).all() and np.allclose(pen[~miss], fr["shadow"][~miss, 0].astype(np.float64) * 0.00465, rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("which", BACKENDS)
def test_pack_then_unpack_round_trip(request, pkg, api, which):
    """producer -> consumer without a denoiser in between: linear radiance survives REBLUR's YCoCg fp16 texel to fp16 precision"""
    from nrd_sample_amd import sample_passes as sp

    backend = request.getfixturevalue(which)
    w, h = 64, 32
    fr = random_frame(w * h, 8)
    fr["diff"][::97, 0] = 2.0
    fr["spec"][::89, 3] = 1.0
    got = run_pack(pkg, backend, fr, w, h, 0, False)
    import torch

    def dev(a):
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a).to(backend.device) if backend.is_device else a

    od = dev(np.zeros((h, w * 4), np.int16))
    sp.backend_unpack(backend, w, h, diff=dev(got["out_diff"]), out_diff=od)
    if backend.is_device:
        torch.cuda.synchronize()
        od = od.cpu().numpy()
    rgb = od.view(np.float16).reshape(h * w, 4).astype(np.float64)
    want = fr["diff"][:, :3].astype(np.float64)
    assert np.allclose(rgb[:, :3], want, rtol=4e-3, atol=2e-3 * want.max(1, keepdims=True))


def compose_reference(fr, w, h, v2w, frustum, sh, relax, hair):
    """numpy float64 restatement of nrdhip_compose (include/nrd_frontend.h: NRD_SG_ReJitter, NRD_MaterialFactors)"""
    f8 = np.float64
    nr = fr["nr"].reshape(h, w)
    ox, oy = (nr & 1023) / 1023.0, ((nr >> 10) & 1023) / 1023.0
    fx, fy = ox * 2 - 1, oy * 2 - 1
    nz = 1 - np.abs(fx) - np.abs(fy)
    t = np.clip(-nz, 0, 1)
    N = np.stack([fx + np.where(fx >= 0, -t, t), fy + np.where(fy >= 0, -t, t), nz], -1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    rough = ((nr >> 20) & 1023) / 1023.0
    mat = nr >> 30
    yy, xx = np.mgrid[0:h, 0:w]
    u, v = (xx + 0.5) / w, (yy + 0.5) / h
    Xv = np.stack([u * frustum[2] + frustum[0], v * frustum[3] + frustum[1], np.ones_like(u)], -1)
    V = (-Xv) @ np.asarray(v2w, f8).reshape(3, 3).T
    V /= np.linalg.norm(V, axis=-1, keepdims=True)

    def sg(sh0, sh1):
        sh0, sh1 = sh0.astype(f8), sh1.astype(f8)
        c0 = (0.25 * sh0[..., 0] + 0.5 * sh0[..., 1] + 0.25 * sh0[..., 2]) if relax else sh0[..., 0]
        return c0, sh1[..., :3]

    def scale(c0, c1, d):
        return np.maximum(0.5 * c0 + (d * c1).sum(-1), 0) * (2 / 3) / np.maximum(c0, 1e-6)

    def specdir(Nq, r):
        nov = (Nq * V).sum(-1, keepdims=True)
        R = Nq * 2 * nov - V
        s = np.clip(1 - r, 0, 1)
        fdom = (s * (np.sqrt(s) + r))[..., None]
        d = Nq + (R - Nq) * fdom
        return d / np.linalg.norm(d, axis=-1, keepdims=True)

    diff, spec = fr["diff"].astype(f8).reshape(h, w, 4), fr["spec"].astype(f8).reshape(h, w, 4)
    if sh:
        dc0, dc1 = sg(fr["dsh0"].reshape(h, w, 4), fr["dsh1"].reshape(h, w, 4))
        sc0, sc1 = sg(fr["ssh0"].reshape(h, w, 4), fr["ssh1"].reshape(h, w, 4))
        Z = fr["viewz"].reshape(h, w).astype(f8)
        dcn, scn = scale(dc0, dc1, N), scale(sc0, sc1, specdir(N, rough))
        dsum, ssum, wsum = dcn.copy(), scn.copy(), np.ones_like(dcn)
        for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            px, py = np.clip(xx + dx, 0, w - 1), np.clip(yy + dy, 0, h - 1)
            Nq, Zq = N[py, px], Z[py, px]
            wgt = (np.abs(Zq - Z) <= 0.05 * np.maximum(np.abs(Z), np.abs(Zq))).astype(f8)
            dsum += scale(dc0, dc1, Nq) * wgt
            ssum += scale(sc0, sc1, specdir(Nq, rough)) * wgt
            wsum += wgt
        dm, sm = dsum / wsum, ssum / wsum
        diff[..., :3] *= np.where(dm > 1e-6, np.minimum(dcn / np.maximum(dm, 1e-30), 2), 1)[..., None]
        spec[..., :3] *= np.where(sm > 1e-6, np.minimum(scn / np.maximum(sm, 1e-30), 2), 1)[..., None]
    b = fr["bcm"].reshape(h, w)
    srgb = np.stack([(b & 255), (b >> 8) & 255, (b >> 16) & 255], -1) / 255.0
    base = np.where(srgb <= 0.04045, srgb / 12.92, ((srgb + 0.055) / 1.055) ** 2.4)
    metal = ((b >> 24) / 255.0)[..., None]
    albedo, rf0 = base * (1 - metal), 0.04 + (base - 0.04) * metal
    nov = np.abs((N * V).sum(-1))
    m = rough * rough
    X1, X2, X3, Y1, Y2, Y3 = nov, nov * nov, nov ** 3, m, m * m, m ** 3
    bias = ((0.99044 - 1.28514 * X1) + (1.29678 - 0.755907 * X1) * Y1) / ((1 + 2.92338 * X1 + 59.4188 * X3) + (20.3225 - 27.0302 * X1 + 222.592 * X3) * Y1 + (121.563 + 626.13 * X1 + 316.627 * X3) * Y3)
    sc = ((0.0365463 + 3.32707 * X1) + (9.0632 - 9.04756 * X1) * Y1) / ((1 + 3.59685 * X2 - 1.36772 * X3) + (9.04401 - 16.3174 * X2 + 9.22949 * X3) * Y1 + (5.56589 + 19.7886 * X2 - 20.2123 * X3) * Y3)
    fenv = np.clip(rf0 * sc[..., None] + bias[..., None], 0, 1)
    dfac, sfac = (1 - fenv) * albedo * 0.99 + 0.01, fenv * 0.99 + 0.01
    keep = (mat == hair)[..., None]
    dfac, sfac = np.where(keep, 1.0, dfac), np.where(keep, 1.0, sfac)
    diff[..., :3] *= dfac
    spec[..., :3] *= sfac
    return diff.astype(np.float16), spec.astype(np.float16)


@pytest.mark.parametrize("which", BACKENDS)
@pytest.mark.parametrize("sh,relax", [(False, False), (True, False), (True, True)])
def test_compose_against_numpy_float64(request, pkg, which, sh, relax):
    import torch

    from nrd_sample_amd import sample_passes as sp, synth

    backend = request.getfixturevalue(which)
    w, h = 80, 48
    n = w * h
    rng = np.random.default_rng(17 + sh + 2 * relax)
    yy, xx = np.mgrid[0:h, 0:w]
    nrm = np.stack([0.3 * np.sin(xx * 0.4), 0.3 * np.cos(yy * 0.5), -np.ones((h, w))], -1) + 0.05 * rng.standard_normal((h, w, 3))
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    fr = {}
    fr["nr"] = synth.pack_normal_roughness(nrm, rng.random((h, w)), rng.integers(0, 4, (h, w))).astype(np.uint32).reshape(-1)
    fr["viewz"] = np.where(xx < w // 2, 5.0, 9.0).astype(np.float32) + 0.01 * rng.standard_normal((h, w)).astype(np.float32)
    h4 = lambda lo, hi: (rng.random((n, 4)) * (hi - lo) + lo).astype(np.float16)
    fr["diff"], fr["spec"] = h4(0, 4), h4(0, 4)
    fr["dsh0"], fr["ssh0"] = h4(0.1, 3), h4(0.1, 3)
    fr["dsh1"], fr["ssh1"] = h4(-1, 1), h4(-1, 1)
    fr["bcm"] = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    v2w = np.array([[0.8, 0, 0.6], [0, 1, 0], [-0.6, 0, 0.8]], np.float32)
    frustum = (-1.0, 0.6, 2.0, -1.2)
    want_d, want_s = compose_reference(fr, w, h, v2w, frustum, sh, relax, hair=2)

    def dev(a, ch, dt):
        a = np.ascontiguousarray(a).view(dt).reshape(h, w * ch)
        return torch.from_numpy(a.copy()).to(backend.device) if backend.is_device else a.copy()

    od, os_ = dev(np.zeros((n, 4), np.float16), 4, np.int16), dev(np.zeros((n, 4), np.float16), 4, np.int16)
    planes = dict(diff=dev(fr["diff"], 4, np.int16), spec=dev(fr["spec"], 4, np.int16), normal_roughness=dev(fr["nr"], 1, np.int32), viewz=dev(fr["viewz"], 1, np.float32),
                  base_color_metalness=dev(fr["bcm"], 1, np.int32), out_diff=od, out_spec=os_)
    if sh:
        planes.update(diff_sh0=dev(fr["dsh0"], 4, np.int16), diff_sh1=dev(fr["dsh1"], 4, np.int16), spec_sh0=dev(fr["ssh0"], 4, np.int16), spec_sh1=dev(fr["ssh1"], 4, np.int16))
    sp.compose(backend, w, h, sh=sh, relax=relax, hair_material_id=2, view_to_world=v2w, camera_frustum=frustum, **planes)
    if backend.is_device:
        torch.cuda.synchronize()
        od, os_ = od.cpu().numpy(), os_.cpu().numpy()
    for got, want, name in ((od, want_d, "diff"), (os_, want_s, "spec")):
        d = ulp16(got.view(np.float16).reshape(n, 4), want.reshape(n, 4))
        assert d.max() <= 2 and (d > 1).mean() < 1e-3, (name, int(d.max()), float((d > 0).mean()))
    hair = (fr["nr"] >> 30) == 2
    if not sh:  # hair keeps factor 1: the radiance passes through untouched
        assert np.array_equal(od.view(np.uint16).reshape(n, 4)[hair], fr["diff"].view(np.uint16).reshape(n, 4)[hair])
