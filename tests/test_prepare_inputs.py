"""PrepareInputs pass: checkerboard resolve (CheckerboardMode::WHITE / BLACK, the sample's default RESOLUTION_HALF tracing,
Source/NRDSample.cpp:267, :545-548, Shaders/TraceOpaque.cs.hlsl:482-508) and hit distance reconstruction
(HitDistanceReconstructionMode::AREA_3X3 / AREA_5X5, :548, UI :1467-1471)."""
import numpy as np
import pytest

import util


def hooks(pkg, api, checker, holes, seed=3):
    """frame hook: punch holes into the hit distances (w = 0) and/or checkerboard the noisy inputs"""
    rng = np.random.default_rng(seed)

    def hook(f, fr):
        if holes:
            for key in ("diff", "spec", "diff_dirocc"):
                if key in fr:
                    a = np.array(fr[key])
                    m = rng.random(a.shape[:2]) < 0.4
                    a[m, 3] = 0
                    fr[key] = a
            for key in ("diff_hitdist", "spec_hitdist"):
                if key in fr:
                    a = np.array(fr[key])
                    a[rng.random(a.shape[:2]) < 0.4] = 0
                    fr[key] = a
        if checker:
            fr.update(pkg.harness.to_checkerboard(fr, f, white=(checker == "WHITE")))
    return hook


def settings(api, scene, dd, checker, recon):
    kw = dict(minMaterialForDiffuse=0, minMaterialForSpecular=1)
    st = util.default_settings(api, scene, dd, **kw)
    for d in dd:
        if d.name.startswith("REBLUR") or d.name.startswith("RELAX"):
            st[d].checkerboardMode = int(getattr(api.CheckerboardMode, checker or "OFF"))
            st[d].hitDistanceReconstructionMode = int(getattr(api.HitDistanceReconstructionMode, recon or "OFF"))
    return st


def test_to_checkerboard_layout(pkg):
    h, w = 6, 9
    a = np.arange(h * w, dtype=np.float32).reshape(h, w, 1).repeat(4, 2).astype(np.float16)
    for f in (0, 1):
        out = pkg.harness.to_checkerboard({"diff": a, "spec": a}, f, white=True)
        assert out["diff"].shape == (h, 5, 4)
        for y in range(h):
            for x in range(w):
                cb = ((x ^ y) ^ f) & 1
                key = "diff" if cb == 1 else "spec"
                assert out[key][y, x >> 1, 0] == a[y, x, 0]


def test_dispatch_list_gains_prepare_pass(pkg, api, oracle):
    D = api.Denoiser
    scene = pkg.synth.Scene(64, 48)
    for separate, prepass, total in ((True, "REBLUR::PrePass", 8), (False, "REBLUR::PrePassTemporalAccumulation", 7)):
        hz = pkg.harness.Harness(oracle, [D.REBLUR_DIFFUSE_SPECULAR], 64, 48, separate_passes=separate)
        fr = scene.frame(0)
        hz.nrd.set_common_settings(scene.common_settings(api, fr, 0, reset=True))
        assert [x["name"] for x in hz.nrd.dispatches([int(D.REBLUR_DIFFUSE_SPECULAR)])][:2] == ["REBLUR::ClassifyTiles", prepass]
        s = api.ReblurSettings(checkerboardMode=int(api.CheckerboardMode.WHITE))
        hz.nrd.set_denoiser_settings(int(D.REBLUR_DIFFUSE_SPECULAR), s)
        names = [x["name"] for x in hz.nrd.dispatches([int(D.REBLUR_DIFFUSE_SPECULAR)])]
        assert names[:3] == ["REBLUR::ClassifyTiles", "REBLUR::PrepareInputs", prepass] and len(names) == total


def test_checkerboard_fixed_point(pkg, api, oracle):
    """constant radiance on a flat plane, checkerboarded: resolve + the whole pipeline still return the constant"""
    D = api.Denoiser
    w, h = 48, 32
    dens = [D.REBLUR_DIFFUSE_SPECULAR]
    hz = pkg.harness.Harness(oracle, dens, w, h)
    st = {dens[0]: api.ReblurSettings(checkerboardMode=int(api.CheckerboardMode.WHITE), hitDistanceReconstructionMode=int(api.HitDistanceReconstructionMode.AREA_3X3))}
    for f in range(4):
        fr = util.flat_frame(pkg, w, h)
        ref = {k: fr[k].copy() for k in ("diff", "spec")}
        fr["diff"][::3, ::2, 3] = 0  # holes in the hit distance
        fr.update(pkg.harness.to_checkerboard(fr, f, white=True))
        hz.frame(util.static_common(api, w, h, frame_index=f, reset=(f == 0)), hz.upload(fr), st)
        for key, out in (("diff", "out_diff"), ("spec", "out_spec")):
            assert util.max_ulp_f16(hz.output(out), ref[key]) <= 2, (f, key)


def test_hit_distance_reconstruction_fills_holes(pkg, api, oracle):
    D = api.Denoiser
    w, h = 40, 24
    dens = [D.REBLUR_DIFFUSE]
    for mode, frac in (("AREA_3X3", 0.5), ("AREA_5X5", 0.9)):
        hz = pkg.harness.Harness(oracle, dens, w, h)
        st = {dens[0]: api.ReblurSettings(hitDistanceReconstructionMode=int(getattr(api.HitDistanceReconstructionMode, mode)))}
        fr = util.flat_frame(pkg, w, h, norm_hit=0.25)
        rng = np.random.default_rng(1)
        holes = rng.random((h, w)) < frac
        holes[h // 2, w // 2] = False
        fr["diff"][holes, 3] = 0
        hz.frame(util.static_common(api, w, h, reset=True), hz.upload(fr), st)
        prep = hz.pool("REBLUR::Prepared_Diff").view(np.float16).reshape(h, w, 4)
        filled = prep[..., 3].astype(np.float32)
        # every hole that had at least one valid neighbour in range is now the constant; rgb untouched
        assert np.array_equal(prep[..., :3], fr["diff"][..., :3])
        assert np.all((np.abs(filled - 0.25) < 1e-3) | (filled == 0))
        assert (filled == 0).sum() < holes.sum() * (0.1 if mode == "AREA_3X3" else 0.35)


@pytest.mark.parametrize("dens,checker,recon", [
    (["REBLUR_DIFFUSE_SPECULAR"], "WHITE", "AREA_5X5"), (["REBLUR_DIFFUSE"], "BLACK", None), (["REBLUR_SPECULAR"], None, "AREA_3X3"),
    (["RELAX_DIFFUSE_SPECULAR_SH"], "WHITE", "AREA_3X3"), (["REBLUR_DIFFUSE_SPECULAR_OCCLUSION"], "WHITE", "AREA_3X3"),
    (["REBLUR_DIFFUSE_SPECULAR_SH"], "BLACK", None), (["REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION"], "WHITE", "AREA_3X3")])
def test_prepare_inputs_emulated_bit_exact(pkg, api, oracle, emulated, dens, checker, recon):
    w, h = 56, 40
    scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR")
    dd = [api.Denoiser[x] for x in dens]
    st = settings(api, scene, dd, checker, recon)
    ho = util.run_frames(api, pkg.harness, oracle, scene, dd, 2, settings=st, frame_hook=hooks(pkg, api, checker, recon is not None))
    he = util.run_frames(api, pkg.harness, emulated, scene, dd, 2, settings=st, frame_hook=hooks(pkg, api, checker, recon is not None))
    assert util.compare_all(ho, he, exact=True) == []


@pytest.mark.gpu
@pytest.mark.parametrize("dens,checker,recon", [
    (["REBLUR_DIFFUSE_SPECULAR"], "WHITE", "AREA_5X5"), (["REBLUR_DIFFUSE"], "BLACK", "AREA_3X3"),
    (["RELAX_DIFFUSE_SPECULAR_SH"], "WHITE", "AREA_3X3"), (["REBLUR_DIFFUSE_SPECULAR_OCCLUSION"], "WHITE", "AREA_5X5"),
    (["RELAX_DIFFUSE_SPECULAR"], "BLACK", None)])
def test_prepare_inputs_hip_bit_exact(pkg, api, oracle, hip, dens, checker, recon):
    w, h = 250, 141
    scene = pkg.synth.Scene(w, h, dolly=0.03, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR")
    dd = [api.Denoiser[x] for x in dens]
    st = settings(api, scene, dd, checker, recon)
    ho = util.run_frames(api, pkg.harness, oracle, scene, dd, 3, settings=st, frame_hook=hooks(pkg, api, checker, recon is not None))
    hh = util.run_frames(api, pkg.harness, hip, scene, dd, 3, settings=st, frame_hook=hooks(pkg, api, checker, recon is not None))
    assert util.compare_all(ho, hh, exact=True) == []


# ---- RGBA16_SNORM DIRECTIONAL_OCCLUSION planes (the sample's data format without DLSS, Source/NRDSample.cpp:2937) -----------------
def to_snorm16(a):
    """fp16 [H, W, 4] in [-1, 1] -> int16 snorm texels"""
    f = np.clip(np.asarray(a, dtype=np.float32), -1.0, 1.0)
    return np.floor(f * np.float32(32767.0) + np.float32(0.5)).astype(np.int16)


def run_dirocc(pkg, api, backend, snorm, frames=3, w=56, h=40, split=0.0):
    den = api.Denoiser.REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION
    scene = pkg.synth.Scene(w, h, dolly=0.04)
    hz = pkg.harness.Harness(backend, [den], w, h)
    if snorm:
        hz.format_override = {"diff_dirocc": api.Format.RGBA16_SNORM, "out_diff_dirocc": api.Format.RGBA16_SNORM}
    st = settings(api, scene, [den], None, None)
    for f in range(frames):
        fr = scene.frame(f)
        if snorm:
            fr["diff_dirocc"] = to_snorm16(fr["diff_dirocc"])
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        cs.splitScreen = split
        hz.frame(cs, hz.upload(fr), st)
    return hz


def test_dirocc_snorm_planes(pkg, api, oracle, emulated):
    ho, he = run_dirocc(pkg, api, oracle, True), run_dirocc(pkg, api, emulated, True)
    assert util.compare_all(ho, he, exact=True) == []
    # same result as the RGBA16_SFLOAT run up to the quantisation of the two formats
    hf = run_dirocc(pkg, api, oracle, False)
    a = ho.output("out_diff_dirocc", dtype=np.int16).astype(np.float32) / 32767.0
    b = hf.output("out_diff_dirocc").astype(np.float32)
    assert float(np.abs(a - b).max()) < 4e-3
    assert float(np.abs(b).max()) > 0.1
    # split screen passes the snorm input through (via the fp16 prepared planes)
    hs = run_dirocc(pkg, api, oracle, True, frames=1, split=1.0)
    src = to_snorm16(pkg.synth.Scene(56, 40, dolly=0.04).frame(0)["diff_dirocc"]).astype(np.float32) / 32767.0
    out = hs.output("out_diff_dirocc", dtype=np.int16).astype(np.float32) / 32767.0
    z = np.asarray(pkg.synth.Scene(56, 40, dolly=0.04).frame(0)["viewz"], dtype=np.float32)
    assert float(np.abs(out - src)[z < 1e4].max()) < 1e-3


def test_dirocc_format_validation(pkg, api, oracle):
    """RGBA16_SNORM is accepted on the DIRECTION_HITDIST slots only"""
    den = api.Denoiser.REBLUR_DIFFUSE
    w, h = 32, 32
    scene = pkg.synth.Scene(w, h)
    hz = pkg.harness.Harness(oracle, [den], w, h)
    hz.format_override = {"diff": api.Format.RGBA16_SNORM}
    fr = scene.frame(0)
    with pytest.raises(api.NrdError):
        hz.frame(scene.common_settings(api, fr, 0, reset=True), hz.upload(fr), {den: api.ReblurSettings()})


@pytest.mark.gpu
def test_dirocc_snorm_hip_bit_exact(pkg, api, oracle, hip):
    ho, hh = run_dirocc(pkg, api, oracle, True, w=250, h=141), run_dirocc(pkg, api, hip, True, w=250, h=141)
    assert util.compare_all(ho, hh, exact=True) == []
