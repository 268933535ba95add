"""The two build flavours of the library (csrc/nrd_device.h, oracle/orc_math.h NRD_UPSTREAM_FORMULAS). The DEFAULT (libnrdhip.so,
liboracle.so; since round 4) carries the recalled upstream forms of ledger rows 1, 2, 7, 13 (oracle/README.md): hit-distance weight
exp(-3 |x|), normal weight on the angle (upstream's AcosApprox: the chord of the two normals), Blur rotation per pixel, RELAX in linear RGB from input to output. The FROZEN flavour
(libnrdhip_frozen.so, liboracle_frozen.so: -DNRD_UPSTREAM_FORMULAS=0) keeps the cheaper forms rounds 1-3 shipped: (1 - |x|)^2, squared
angle, rotation per 2x2 quad, YCoCg inside RELAX. Same sources; every test of the suite that takes `oracle` / `emulated` / `hip` runs the
default flavour - here the frozen one is held to ITS oracle just as exactly, and the distance between the two is put on record
(bench.py reports the same distance at 1920x1080 after 34 frames: config.frozen_formulas.distance_from_default)."""
import numpy as np
import pytest

import util

CASES = [["REBLUR_DIFFUSE_SPECULAR"], ["RELAX_DIFFUSE_SPECULAR"], ["REBLUR_DIFFUSE_SPECULAR_SH"]]


def run(pkg, api, backend, dens, w, h, frames):
    scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR")
    dd = [api.Denoiser[x] for x in dens]
    st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1)
    return util.run_frames(api, pkg.harness, backend, scene, dd, frames, settings=st)


@pytest.mark.parametrize("dens", CASES)
def test_frozen_flavour_emulated_bit_exact(pkg, api, oracle_frozen, emulated_frozen, dens):
    ho = run(pkg, api, oracle_frozen, dens, 72, 40, 2)
    he = run(pkg, api, emulated_frozen, dens, 72, 40, 2)
    assert util.compare_all(ho, he, exact=True) == []


def test_frozen_flavour_differs_from_the_default(pkg, api, oracle, oracle_frozen):
    """the switch does something: same inputs, different (but close) outputs - and the size of the difference is on record"""
    dens = ["REBLUR_DIFFUSE_SPECULAR"]
    a = run(pkg, api, oracle_frozen, dens, 96, 64, 4)
    b = run(pkg, api, oracle, dens, 96, 64, 4)
    for key in ("out_diff", "out_spec"):
        x, y = a.output(key).astype(np.float32), b.output(key).astype(np.float32)
        assert not np.array_equal(x, y)
        p = util.psnr(x, y)
        print("%s: frozen flavour vs default (recalled upstream formulas) PSNR %.1f dB, %.1f %% of the values differ by more than 1 fp16 ULP" % (
            key, p, 100.0 * float((np.abs(util.f16_ordered(a.output(key)) - util.f16_ordered(b.output(key))) > 1).mean())))
        assert 25.0 < p < 80.0


def test_relax_stays_in_linear_rgb(pkg, api, oracle, oracle_frozen, emulated):
    """ledger row 13 (default flavour): with anti-firefly and fast-history clamping on (the luminance clamps that must scale r, g and b
    alike) the emulated kernels match the oracle bit for bit; against the frozen YCoCg-inside build the outputs differ, by how much is
    printed; and a GREY input stays grey to the last bit - in linear RGB no chroma channel exists that rounding could tint"""
    dens = ["RELAX_DIFFUSE_SPECULAR"]
    scene = pkg.synth.Scene(72, 40, dolly=0.04, denoiser="RELAX")
    dd = [api.Denoiser[x] for x in dens]
    st = {dd[0]: api.RelaxSettings(enableAntiFirefly=True, minMaterialForDiffuse=0, minMaterialForSpecular=1)}
    ho = util.run_frames(api, pkg.harness, oracle, scene, dd, 3, settings=st)
    he = util.run_frames(api, pkg.harness, emulated, scene, dd, 3, settings=st)
    assert util.compare_all(ho, he, exact=True) == []
    hf = util.run_frames(api, pkg.harness, oracle_frozen, scene, dd, 3, settings=st)
    for key in ("out_diff", "out_spec"):
        x, y = hf.output(key).astype(np.float32), ho.output(key).astype(np.float32)
        assert not np.array_equal(x, y)
        print("%s: RELAX YCoCg-inside (frozen flavour) vs linear-RGB (default) PSNR %.1f dB" % (key, util.psnr(x, y)))
        assert util.psnr(x, y) > 25.0

    def grey(f, fr):
        for key in ("diff", "spec"):
            v = fr[key].copy()
            v[..., 1] = v[..., 0]
            v[..., 2] = v[..., 0]
            fr[key] = v

    hg = util.run_frames(api, pkg.harness, oracle, scene, dd, 3, settings=st, frame_hook=grey)
    for key in ("out_diff", "out_spec"):
        o = hg.output(key)
        assert np.array_equal(o[..., 0], o[..., 1]) and np.array_equal(o[..., 0], o[..., 2]), key


@pytest.mark.gpu
@pytest.mark.parametrize("dens", CASES)
def test_frozen_flavour_hip_matches_its_oracle(pkg, api, oracle_frozen, hip_frozen, dens):
    w, h = 480, 270
    scene = pkg.synth.Scene(w, h, dolly=0.02, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR")
    dd = [api.Denoiser[x] for x in dens]
    st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1)
    ho = pkg.harness.Harness(oracle_frozen, dd, w, h)
    oracle_frozen.lib.orc_set_threads(ho.nrd.handle, 16)
    hg = pkg.harness.Harness(hip_frozen, dd, w, h)
    for f in range(4):
        fr = scene.frame(f)
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        ho.frame(cs, ho.upload(fr), st)
        hg.frame(cs, hg.upload(fr), st)
        assert util.compare_all(ho, hg, exact=True) == [], "frame %d" % f
