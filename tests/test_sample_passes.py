"""ConfidenceBlur producer (Shaders/ConfidenceBlur.cs.hlsl, Source/NRDSample.cpp:3999-4026) and the NRD back-end unpack
(Shaders/Composition.cs.hlsl:57-64, 74-175): oracle known answers, emulated-kernel parity on CPU, HIP parity on the GPU."""
import numpy as np
import pytest


def load_sp(pkg):
    import importlib
    return importlib.import_module("nrd_sample_amd.sample_passes")


FRUSTUM = (-1.0, 0.5625, 2.0, -1.125)  # x0, y0, dx, dy for a 90 deg hFOV 16:9 camera


def run_conf(sp, backend, grad, relax=False, frame=3, passes=5, to_dev=None):
    h, w = grad.shape[:2]
    ping = np.ascontiguousarray(grad).view(np.uint8).reshape(h, w * 8).copy()
    pong = np.zeros_like(ping)
    if to_dev is not None:
        ping, pong = to_dev(ping), to_dev(pong)
    sp.confidence_blur(backend, ping, pong, w, h, FRUSTUM, rect_width=w * 5, unproject=1.0 / (0.5 * h * 5 * 1.7777), frame_index=frame,
                       max_accumulated_frame_num=60, relax=relax, passes_num=passes)
    if to_dev is not None:
        ping, pong = ping.cpu().numpy(), pong.cpu().numpy()
    return ping.view(np.float16).reshape(h, w, 4), pong.view(np.float16).reshape(h, w, 4)


def flat_gradient(w, h, g):
    out = np.zeros((h, w, 4), np.float16)
    out[..., 0] = g
    out[..., 1:3] = 0.5  # octahedral (0.5, 0.5) = +z normal
    out[..., 3] = 10.0 * 0.125
    return out


def test_confidence_fixed_point_and_mapping(pkg, oracle):
    sp = load_sp(pkg)
    g = flat_gradient(48, 32, 0.25)
    ping, pong = run_conf(sp, oracle, g, passes=4)  # 4 blur passes of a constant on a fronto-parallel plane: still the constant
    assert np.all(ping[..., 0] == np.float16(0.25))  # pass 4 (index 3) writes ping
    _, pong = run_conf(sp, oracle, g, passes=5)
    # last pass: confidence = 1 - sRGB(uncharted(g)) + dither / maxAccum, in [0, 1]; yzw untouched
    x = 0.25
    A, B, C_, D, E, F = 0.22, 0.3, 0.1, 0.2, 0.01, 0.3
    f = lambda v: (v * (A * v + C_ * B) + D * E) / (v * (A * v + B) + D * F) - E / F
    lin = f(x) / f(11.2)
    srgb = 1.055 * lin ** (1 / 2.4) - 0.055
    want = 1.0 - srgb
    assert np.all(np.abs(pong[..., 0].astype(np.float32) - want) <= 0.5 / 60 + 2e-3)
    assert np.array_equal(pong[..., 1:].view(np.uint16), g[..., 1:].view(np.uint16))
    # dither pattern has period 4 in x and y
    assert np.array_equal(pong[4:8, 4:8, 0], pong[8:12, 12:16, 0])
    # RELAX squares the confidence
    _, pr = run_conf(sp, oracle, g, relax=True)
    assert float(pr[..., 0].astype(np.float32).mean()) < float(pong[..., 0].astype(np.float32).mean())


def test_confidence_monotonic_and_sky(pkg, oracle):
    sp = load_sp(pkg)
    lo, hi = flat_gradient(32, 32, 0.05), flat_gradient(32, 32, 2.0)
    _, a = run_conf(sp, oracle, lo)
    _, b = run_conf(sp, oracle, hi)
    assert float(a[..., 0].astype(np.float32).min()) > float(b[..., 0].astype(np.float32).max())  # larger gradient -> less confidence
    g = flat_gradient(32, 32, 0.5)
    g[5, 7, 3] = 2.0e5 * 0.125  # sky texel: |z| > INF
    g[5, 7, 0] = 123.0
    _, p = run_conf(sp, oracle, g)
    assert p[5, 7, 0] == np.float16(1.0) and p[5, 7, 3] == g[5, 7, 3]
    ping, _ = run_conf(sp, oracle, g, passes=4)
    assert ping[5, 7, 0] == np.float16(0.0)
    # the far-away sky texel never leaks into its neighbours (plane-distance weight)
    assert np.all(np.abs(ping[4:7, 6:9, 0].astype(np.float32)[ping[4:7, 6:9, 3] < 100] - 0.5) < 1e-3)


def test_confidence_normal_edge_stops_blur(pkg, oracle):
    sp = load_sp(pkg)
    g = flat_gradient(64, 32, 0.0)
    g[:, 32:, 0] = 1.0
    g[:, 32:, 1] = 1.0  # right half: octahedral (1.0, 0.5) = +x normal, perpendicular to the left half's +z
    ping, _ = run_conf(sp, oracle, g, passes=4)
    assert np.all(ping[:, :32, 0] == 0) and np.all(ping[:, 32:, 0] == 1)


@pytest.mark.parametrize("relax,size", [(False, (80, 48)), (True, (80, 48)), (False, (53, 37))])
def test_confidence_emulated_bit_exact(pkg, oracle, emulated, relax, size):
    """(53 x 37: ragged tiles - the LDS window of a border tile holds positions outside the frame: unclamped uv, clamped texel)"""
    sp = load_sp(pkg)
    g = sp.synth_gradient(size[0], size[1], seed=5)
    po, qo = run_conf(sp, oracle, g, relax=relax)
    pe, qe = run_conf(sp, emulated, g, relax=relax)
    assert np.array_equal(po.view(np.uint16), pe.view(np.uint16)) and np.array_equal(qo.view(np.uint16), qe.view(np.uint16))


def make_unpack_inputs(w, h, seed):
    rng = np.random.default_rng(seed)
    mk = lambda: rng.random((h, w, 4)).astype(np.float16)
    d = dict(diff=mk(), spec=mk(), diff_sh1=(mk() - np.float16(0.5)), spec_sh1=(mk() - np.float16(0.5)))
    d["nr"] = rng.integers(0, 2 ** 32, (h, w), dtype=np.uint32)
    d["shadow"] = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    d["occ"] = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    return d


def run_unpack(sp, api, backend, d, mode, relax, resolve, to_dev=None):
    h, w = d["nr"].shape
    b = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(h, -1).copy()
    dev = (lambda a: a) if to_dev is None else to_dev
    outs = [dev(np.zeros((h, w * 8), np.uint8)) for _ in range(3)]
    occ = mode == api.UNPACK_OCCLUSION
    sp.backend_unpack(backend, w, h, mode=mode, relax=relax, resolve=resolve,
                      diff=dev(b(d["occ"] if occ else d["diff"])), spec=dev(b(d["occ"] if occ else d["spec"])),
                      diff_sh1=dev(b(d["diff_sh1"])), spec_sh1=dev(b(d["spec_sh1"])), normal_roughness=dev(b(d["nr"])),
                      shadow=dev(b(d["shadow"])), out_diff=outs[0], out_spec=outs[1], out_shadow=outs[2],
                      view_to_world=np.eye(3), camera_frustum=FRUSTUM)
    return [(o.cpu().numpy() if hasattr(o, "cpu") else o).view(np.uint16) for o in outs]


def test_unpack_known_answers(pkg, api, oracle):
    sp = load_sp(pkg)
    d = make_unpack_inputs(24, 16, 1)
    # REBLUR NORMAL: YCoCg -> linear; the hit distance channel passes through; shadow is stored sqrt-encoded
    rgb = np.array([0.8, 0.4, 0.2], np.float32)
    y, co, cg = 0.25 * rgb[0] + 0.5 * rgb[1] + 0.25 * rgb[2], 0.5 * rgb[0] - 0.5 * rgb[2], -0.25 * rgb[0] + 0.5 * rgb[1] - 0.25 * rgb[2]
    d["diff"][...] = np.array([y, co, cg, 0.3], np.float16)
    d["shadow"][...] = 255
    d["shadow"][0, 0] = (0, 128, 255, 64)
    od, os_, osh = run_unpack(sp, api, oracle, d, api.UNPACK_NORMAL, False, False)
    got = od.view(np.float16).reshape(16, 24, 4).astype(np.float32)
    assert np.allclose(got[..., :3], rgb, atol=2e-3) and np.allclose(got[..., 3], 0.3, atol=1e-3)
    sh = osh.view(np.float16).reshape(16, 24, 4).astype(np.float32)
    assert np.all(sh[1:] == 1.0) and np.allclose(sh[0, 0], [0.0, (128 / 255) ** 2, 1.0, (64 / 255) ** 2], atol=1e-3)
    # RELAX: identity on rgb, w = 1 / pi
    od, _, _ = run_unpack(sp, api, oracle, d, api.UNPACK_NORMAL, True, False)
    got = od.view(np.float16).reshape(16, 24, 4)
    assert np.array_equal(got[..., :3], d["diff"][..., :3]) and np.allclose(got[..., 3].astype(np.float32), 1 / np.pi, atol=1e-3)
    # OCCLUSION: R16_UNORM hit distance replicated
    d["occ"][...] = 65535
    od, _, _ = run_unpack(sp, api, oracle, d, api.UNPACK_OCCLUSION, False, False)
    assert np.all(od.view(np.float16) == np.float16(1.0))
    # SH resolve: light arriving along the normal keeps its luminance (SH1 = N * Y), light from behind resolves to 0
    d["nr"][...] = 511 | (511 << 10) | (1023 << 20)  # octahedral (0.5, 0.5) ~ +z, roughness 1
    d["diff"][...] = np.array([0.5, 0.0, 0.0, 0.2], np.float16)
    d["diff_sh1"][...] = np.array([0.0, 0.0, 0.5, 0.0], np.float16)
    od, _, _ = run_unpack(sp, api, oracle, d, api.UNPACK_SH, False, True)
    got = od.view(np.float16).reshape(16, 24, 4).astype(np.float32)
    assert np.allclose(got[..., :3], 0.5, atol=5e-3)
    d["diff_sh1"][...] = np.array([0.0, 0.0, -0.5, 0.0], np.float16)
    od, _, _ = run_unpack(sp, api, oracle, d, api.UNPACK_SH, False, True)
    assert np.all(od.view(np.float16).reshape(16, 24, 4)[..., :3] == 0)


@pytest.mark.parametrize("mode,relax,resolve", [(0, False, False), (0, True, False), (1, False, False), (2, False, True), (2, True, True), (2, False, False)])
def test_unpack_emulated_bit_exact(pkg, api, oracle, emulated, mode, relax, resolve):
    sp = load_sp(pkg)
    d = make_unpack_inputs(70, 21, 2)
    a = run_unpack(sp, api, oracle, d, mode, relax, resolve)
    b = run_unpack(sp, api, emulated, d, mode, relax, resolve)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_sample_pass_argument_checks(pkg, api, oracle, emulated):
    import ctypes as C
    for b in (oracle, emulated):
        d = api.ConfidenceBlurDesc()
        assert b.confidence_blur(C.byref(d), None) == int(api.Result.INVALID_ARGUMENT)
        u = api.UnpackDesc()
        assert b.backend_unpack(C.byref(u), None) == int(api.Result.INVALID_ARGUMENT)
        u.width, u.height, u.mode = 8, 8, 7
        assert b.backend_unpack(C.byref(u), None) == int(api.Result.INVALID_ARGUMENT)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("relax", [False, True])
def test_confidence_hip_bit_exact_and_feeds_reblur(pkg, api, oracle, hip, relax):
    import torch
    sp = load_sp(pkg)
    w, h = sp.sharc_dims(1920, 1080)
    g = sp.synth_gradient(w, h, seed=9)
    to_dev = lambda a: torch.from_numpy(a).to("cuda:0")
    po, qo = run_conf(sp, oracle, g, relax=relax)
    ph, qh = run_conf(sp, hip, g, relax=relax, to_dev=to_dev)
    assert np.array_equal(po.view(np.uint16), ph.view(np.uint16)) and np.array_equal(qo.view(np.uint16), qh.view(np.uint16))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,relax,resolve", [(0, False, False), (0, True, False), (1, False, False), (2, False, True), (2, True, True)])
def test_unpack_hip_bit_exact(pkg, api, oracle, hip, mode, relax, resolve):
    import torch
    sp = load_sp(pkg)
    d = make_unpack_inputs(333, 77, 4)
    a = run_unpack(sp, api, oracle, d, mode, relax, resolve)
    b = run_unpack(sp, api, hip, d, mode, relax, resolve, to_dev=lambda t: torch.from_numpy(t).to("cuda:0"))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


# ---------------------------------------------------------------------------------------------------------------------
# TAA (Shaders/Taa.cs.hlsl)
def make_taa_inputs(w, h, seed, motion=(0.0, 0.0)):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([0.5 + 0.4 * np.sin(x * 0.21), 0.5 + 0.4 * np.cos(y * 0.17), 0.3 + 0.2 * np.sin((x + y) * 0.11)], -1)
    comp = np.concatenate([img * (1 + 0.1 * rng.standard_normal((h, w, 1))), np.ones((h, w, 1))], -1).astype(np.float16)
    mv = np.zeros((h, w, 4), np.float16)
    mv[..., 0], mv[..., 1] = motion
    z = np.where(x < w / 2, 4.0, 9.0) * 0.125
    mv[..., 3] = z * np.where((y > h * 0.7), -1.0, 1.0)  # negative w: thin / noisy geometry asks for the 5x5 window
    hist = np.concatenate([img, np.full((h, w, 1), 0.3)], -1).astype(np.float16)
    return mv, comp, hist


def run_taa(sp, backend, mv, comp, hist, to_dev=None, **kw):
    h, w = mv.shape[:2]
    b = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(a.shape[0], -1).copy()
    dev = (lambda a: a) if to_dev is None else to_dev
    out = dev(np.zeros((h, w * 8), np.uint8))
    sp.taa(backend, dev(b(mv)), dev(b(comp)), dev(b(hist)), out, w, h, render_width=hist.shape[1], render_height=hist.shape[0], **kw)
    out = out.cpu().numpy() if hasattr(out, "cpu") else out
    return out.view(np.float16).reshape(h, w, 4)


def test_taa_known_answers(pkg, oracle):
    sp = load_sp(pkg)
    w, h = 64, 40
    # a constant image with a constant history is a fixed point (tonemap off): result = image, mix rate decays as m / (1 + m)
    mv, comp, hist = make_taa_inputs(w, h, 0)
    comp[..., :3] = np.float16(0.4)
    hist[..., :3] = np.float16(0.4)
    hist[..., 3] = np.float16(0.5)
    out = run_taa(sp, oracle, mv, comp, hist, tonemap=False, taa_min_mix=0.05).astype(np.float32)
    assert np.allclose(out[..., :3], 0.4, atol=1e-3)
    assert np.allclose(out[..., 3], 0.5 / 1.5, atol=2e-3)
    # motion pointing outside the screen: no history -> the (tonemapped) input passes through with mix rate 1
    mv2, comp2, hist2 = make_taa_inputs(w, h, 1, motion=(-3.0 * w, 0.0))
    out = run_taa(sp, oracle, mv2, comp2, hist2, tonemap=False).astype(np.float32)
    assert np.allclose(out[..., :3], comp2[..., :3].astype(np.float32), atol=2e-3) and np.all(out[..., 3] == 1.0)
    # a history far outside the neighbourhood colour box is clipped back to it and mostly rejected
    mv3, comp3, hist3 = make_taa_inputs(w, h, 2)
    hist3[..., :3] = np.float16(5.0)
    out = run_taa(sp, oracle, mv3, comp3, hist3, tonemap=False).astype(np.float32)
    assert out[..., :3].max() < 1.5 and out[..., 3].mean() > 0.9
    # integer-pixel motion fetches the history exactly one pixel to the side (bicubic weights collapse at f = 0)
    mv4, comp4, hist4 = make_taa_inputs(w, h, 3, motion=(1.0, 0.0))
    comp4[..., :3] = hist4[..., :3]
    shifted = np.roll(hist4, -1, axis=1)
    out = run_taa(sp, oracle, mv4, comp4, hist4, tonemap=False, taa_min_mix=0.0).astype(np.float32)
    ref0 = run_taa(sp, oracle, np.zeros_like(mv4) + mv4 * np.array([0, 0, 0, 1], np.float16), comp4, shifted, tonemap=False, taa_min_mix=0.0).astype(np.float32)
    assert np.allclose(out[:, 2:-3], ref0[:, 2:-3], atol=2e-3)


@pytest.mark.parametrize("tonemap,motion", [(True, (0.4, -0.7)), (False, (2.5, 1.25))])
def test_taa_emulated_bit_exact(pkg, oracle, emulated, tonemap, motion):
    sp = load_sp(pkg)
    mv, comp, hist = make_taa_inputs(70, 37, 5, motion=motion)
    a = run_taa(sp, oracle, mv, comp, hist, tonemap=tonemap, hdr_scale=1.25)
    b = run_taa(sp, emulated, mv, comp, hist, tonemap=tonemap, hdr_scale=1.25)
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))


@pytest.mark.gpu
@pytest.mark.parametrize("tonemap,motion", [(True, (0.4, -0.7)), (False, (2.5, 1.25))])
def test_taa_hip_bit_exact(pkg, oracle, hip, tonemap, motion):
    import torch
    sp = load_sp(pkg)
    mv, comp, hist = make_taa_inputs(500, 281, 6, motion=motion)
    a = run_taa(sp, oracle, mv, comp, hist, tonemap=tonemap, hdr_scale=1.25)
    b = run_taa(sp, hip, mv, comp, hist, to_dev=lambda t: torch.from_numpy(t).to("cuda:0"), tonemap=tonemap, hdr_scale=1.25)
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))
