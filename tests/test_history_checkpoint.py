"""Checkpoint / resume of the denoiser history ("dump / load of the permanent pool for deterministic multi-frame
fixtures"; include/nrdhip.h nrdhip_history_state): an instance restored from the permanent planes + counters of another one after
frame k continues bit-identically - every output and every plane of frames k + 1 ... - also when k is odd (the ping-pong planes then
have swapped roles: the counters carry the parity)."""
import numpy as np
import pytest

import util


def run(pkg, api, backend, dens, tmp_path, w=96, h=64, split=3, total=6):
    scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR")
    dd = [api.Denoiser[x] for x in dens]
    st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1)
    ha = pkg.harness.Harness(backend, dd, w, h)
    for f in range(split):
        fr = scene.frame(f)
        ha.frame(scene.common_settings(api, fr, f, reset=(f == 0)), ha.upload(fr), st)
    path = str(tmp_path / "history.npz")
    ha.nrd.save_history(path)
    hb = pkg.harness.Harness(backend, dd, w, h)  # a fresh instance: zeroed pools, counters at 0
    hb.nrd.load_history(path)
    for f in range(split, total):
        fr = scene.frame(f)
        cs = scene.common_settings(api, fr, f, reset=False)
        ha.frame(cs, ha.upload(fr), st)
        hb.frame(cs, hb.upload(fr), st)
        for key in ha.outputs:
            assert np.array_equal(ha.fetch(ha.outputs[key]), hb.fetch(hb.outputs[key])), (f, key)
    for pa, pb in zip(ha.nrd.pools[0], hb.nrd.pools[0]):
        assert np.array_equal(ha.fetch(pa["buf"]), hb.fetch(pb["buf"])), pa["name"]


@pytest.mark.parametrize("dens", [["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW", "REFERENCE"], ["RELAX_DIFFUSE_SPECULAR"]])
def test_resume_is_bit_identical_oracle(pkg, api, oracle, tmp_path, dens):
    run(pkg, api, oracle, dens, tmp_path)


def test_resume_is_bit_identical_emulated(pkg, api, emulated, tmp_path):
    run(pkg, api, emulated, ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW", "REFERENCE"], tmp_path, w=48, h=48, split=3, total=5)


def test_history_file_of_another_instance_is_refused(pkg, api, oracle, tmp_path):
    D = api.Denoiser
    ha = pkg.harness.Harness(oracle, [D.REBLUR_DIFFUSE], 64, 48)
    path = str(tmp_path / "h.npz")
    ha.nrd.save_history(path)
    hb = pkg.harness.Harness(oracle, [D.REBLUR_DIFFUSE], 80, 48)
    with pytest.raises(ValueError):
        hb.nrd.load_history(path)


@pytest.mark.gpu
def test_resume_is_bit_identical_hip(pkg, api, hip, tmp_path):
    run(pkg, api, hip, ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW", "REFERENCE"], tmp_path, w=320, h=192)
