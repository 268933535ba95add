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
    hz = pkg.harness.Harness(oracle, [D.REBLUR_DIFFUSE_SPECULAR], 64, 48)
    fr = scene.frame(0)
    hz.nrd.set_common_settings(scene.common_settings(api, fr, 0, reset=True))
    assert [x["name"] for x in hz.nrd.dispatches([int(D.REBLUR_DIFFUSE_SPECULAR)])][:2] == ["REBLUR::ClassifyTiles", "REBLUR::PrePass"]
    s = api.ReblurSettings(checkerboardMode=int(api.CheckerboardMode.WHITE))
    hz.nrd.set_denoiser_settings(int(D.REBLUR_DIFFUSE_SPECULAR), s)
    names = [x["name"] for x in hz.nrd.dispatches([int(D.REBLUR_DIFFUSE_SPECULAR)])]
    assert names[:3] == ["REBLUR::ClassifyTiles", "REBLUR::PrepareInputs", "REBLUR::PrePass"] and len(names) == 8


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
