"""Orthographic projections - the sample's "Ortho" camera (Source/NRDSample.cpp:1214 checkbox, :1971 orthoRange, test 221 at :77).
The kernels of that flavour are a second compilation of the same sources (nrd_device.h NRD_ORTHO); checked here the same way
as the perspective ones: known answers on the oracle, bit-exact emulated kernels on CPU, bit-exact HIP on the GPU."""
import numpy as np
import pytest

import util

DENS = [
    ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW_TRANSLUCENCY", "REFERENCE"],
    ["RELAX_DIFFUSE_SPECULAR"],
    ["REBLUR_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW"],
    ["RELAX_DIFFUSE_SPECULAR_SH"],
    ["REBLUR_DIFFUSE_SPECULAR_OCCLUSION"],
]


def ortho_scene(pkg, dens, w=60, h=44, dolly=0.05):
    return pkg.synth.Scene(w, h, dolly=dolly, denoiser="RELAX" if dens[0].name.startswith("RELAX") else "REBLUR", ortho=True)


def settings_for(api, scene, dens):
    return util.default_settings(api, scene, dens, minMaterialForDiffuse=0, minMaterialForSpecular=1)


def test_scene_is_orthographic(pkg, api):
    scene = ortho_scene(pkg, [api.Denoiser.REBLUR_DIFFUSE_SPECULAR])
    fr = scene.frame(1)
    m = np.asarray(fr["view_to_clip"], dtype=np.float32)
    assert m[11] == 0.0 and m[15] == 1.0
    z = np.asarray(fr["viewz"], dtype=np.float32)
    assert np.isfinite(z[z < 1e4]).all() and (z[z < 1e4] > 0).all()
    # lateral camera motion in an orthographic view moves every (static) surface point by the same pixel offset
    mv = np.asarray(fr["mv"], dtype=np.float32)
    hit = z < 1e4
    assert np.ptp(mv[..., 0][hit]) < 0.02 and abs(float(mv[..., 0][hit].mean())) > 0.1


def test_ortho_history_survives_camera_motion(pkg, api, oracle):
    """reprojection, disocclusion test and bilateral weights work in an orthographic view: with the camera moving, accumulated
    output is much less noisy than the first frame, and the validation overlay reports long histories on the surfaces"""
    den = api.Denoiser.REBLUR_DIFFUSE_SPECULAR
    scene = ortho_scene(pkg, [den], 96, 64, dolly=0.03)
    clean = scene.frame(9, noise=False)
    ref = np.asarray(clean["diff"], dtype=np.float32)[..., 0]
    keep = []

    def hook(f, cs):
        cs.enableValidation = True

    hz = util.run_frames(api, pkg.harness, oracle, scene, [den], 10, settings=settings_for(api, scene, [den]), common_hook=hook, keep=keep)
    z = np.asarray(scene.frame(9)["viewz"], dtype=np.float32)
    inner = np.zeros_like(z, dtype=bool)
    inner[8:-8, 8:-8] = True
    m = (z < 1e4) & inner
    first = np.asarray(keep[0]["out_diff"]).view(np.float16).astype(np.float32).reshape(64, 96, 4)[..., 0]
    last = hz.output("out_diff").astype(np.float32)[..., 0]
    ref0 = np.asarray(scene.frame(0, noise=False)["diff"], dtype=np.float32)[..., 0]
    z0 = np.asarray(scene.frame(0)["viewz"], dtype=np.float32)
    m0 = (z0 < 1e4) & inner
    err_first = float(np.abs(first - ref0)[m0].mean())
    err_last = float(np.abs(last - ref)[m].mean())
    assert err_last < 0.6 * err_first, (err_first, err_last)
    val = hz.fetch(hz.outputs["out_validation"]).reshape(64, 96, 4)
    frames = val[..., 0].astype(np.float32) / 255.0 * 63.0
    assert float(np.median(frames[m])) >= 8.0  # 10 frames in: most surface pixels kept their whole history


@pytest.mark.parametrize("dens", DENS, ids=["+".join(d) for d in DENS])
def test_ortho_emulated_bit_exact(pkg, api, oracle, emulated, dens):
    dd = [api.Denoiser[x] for x in dens]
    scene = ortho_scene(pkg, dd)
    ha = util.run_frames(api, pkg.harness, oracle, scene, dd, 3, settings=settings_for(api, scene, dd))
    hb = util.run_frames(api, pkg.harness, emulated, scene, dd, 3, settings=settings_for(api, scene, dd))
    assert util.compare_all(ha, hb, exact=True) == []


@pytest.mark.gpu
@pytest.mark.parametrize("dens", DENS, ids=["+".join(d) for d in DENS])
def test_ortho_hip_bit_exact(pkg, api, oracle, hip, dens):
    dd = [api.Denoiser[x] for x in dens]
    scene = ortho_scene(pkg, dd, 128, 96)
    ha = util.run_frames(api, pkg.harness, oracle, scene, dd, 4, settings=settings_for(api, scene, dd))
    hb = util.run_frames(api, pkg.harness, hip, scene, dd, 4, settings=settings_for(api, scene, dd))
    assert util.compare_all(ha, hb, exact=True) == []


def test_projection_switch_restarts_cleanly(pkg, api, oracle, emulated):
    """the sample toggles Ortho at run time (Source/NRDSample.cpp:1214) and restarts the accumulation (:2142): the first orthographic
    frame still carries the perspective matrix as 'previous' - accepted, and identical to a fresh orthographic start"""
    den = api.Denoiser.REBLUR_DIFFUSE_SPECULAR
    w, h = 60, 44
    persp, orth = pkg.synth.Scene(w, h, dolly=0.05), pkg.synth.Scene(w, h, dolly=0.05, ortho=True)
    outs = []
    for backend in (oracle, emulated):
        hz = pkg.harness.Harness(backend, [den], w, h)
        st = settings_for(api, persp, [den])
        for f in range(2):
            fr = persp.frame(f)
            hz.frame(persp.common_settings(api, fr, f, reset=(f == 0)), hz.upload(fr), st)
        fr = orth.frame(2)
        cs = orth.common_settings(api, fr, 2, reset=True)
        prev = np.asarray(persp.frame(1)["view_to_clip"], dtype=np.float32)
        for i in range(16):
            cs.viewToClipMatrixPrev[i] = float(prev[i])
        hz.frame(cs, hz.upload(fr), st)
        outs.append(hz.output("out_diff").copy())
        fresh = pkg.harness.Harness(backend, [den], w, h)
        fresh.frame(orth.common_settings(api, fr, 2, reset=True), fresh.upload(fr), st)
        assert np.array_equal(outs[-1].view(np.uint16), fresh.output("out_diff").view(np.uint16))
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))
