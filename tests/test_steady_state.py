"""The regime bench.py times - accumulation saturated, narrow radii, antilag and stabilization at full history - against the CPU oracle.

Every other HIP-vs-oracle test runs 2-6 frames from a restart; the headline is timed after 32. Here the bench's own operating point
(bench.py settings_of: maxAccumulatedFrameNum 30, fast 6, stabilized 30, material-aware filtering, clamp sigma 1.5) runs 36 frames of
the moving-camera scene on both sides and the HIP path must equal the oracle BIT FOR BIT - every OUT_* plane and every pool plane - on
each of the frames 31..36 ("quality gates per run ... on the permanent pool after the last frame")."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

FRAMES, FIRST_CHECKED = 36, 30  # frames 31..36 (0-based 30..35) are compared


def bench_settings(api, scene, dens):
    import bench

    return bench.settings_of(api, scene, dens)


def run_pair(pkg, api, oracle, hip, dens, w, h, threads, exact=True, dolly=0.004, frames=FRAMES, first_checked=FIRST_CHECKED):
    # (frames are rendered on the GPU - a 1080p frame takes numpy ~2 s - and copied to the host for the oracle)
    scene = pkg.synth.Scene(w, h, dolly=dolly, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR", device="cuda:0")
    dd = [api.Denoiser[x] for x in dens]
    st = bench_settings(api, scene, dd)
    ho = pkg.harness.Harness(oracle, dd, w, h)
    oracle.lib.orc_set_threads(ho.nrd.handle, threads)
    hg = pkg.harness.Harness(hip, dd, w, h)
    checked = 0
    for f in range(frames):
        fr = scene.frame(f)
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        ho.frame(cs, ho.upload(util.host_frame(fr)), st)
        hg.frame(cs, hg.upload(fr), st)
        if f >= first_checked:
            bad = util.compare_all(ho, hg, exact=exact, ulp=1)
            assert bad == [], "frame %d: %s" % (f + 1, bad)
            checked += 1
    assert checked == frames - first_checked
    return ho, hg


def saturated(hz):
    """share of the geometry pixels' accumulation-speed codes (Data1: u8 quarter-frames per signal, ping-pong planes) within one frame of the cap"""
    best = 0.0
    for name in ("REBLUR::Data1_A", "REBLUR::Data1_B"):
        d1 = np.asarray(hz.pool(name)).view(np.uint8).reshape(-1, 2)
        geo = d1.any(axis=1)
        if geo.any():
            best = max(best, float((d1[geo] >= 4 * (30 - 1)).mean()))
    return best


def test_steady_state_reblur_1080p_against_oracle(pkg, api, oracle, hip):
    """the headline denoiser at 1920x1080, 36 frames: bit-exact on frames 31..36, and the run really is the saturated regime"""
    ho, hg = run_pair(pkg, api, oracle, hip, ["REBLUR_DIFFUSE_SPECULAR"], 1920, 1080, 128)
    sat = saturated(hg)
    print("share of accumulation-speed codes at the cap after %d frames: %.3f" % (FRAMES, sat))
    assert sat > 0.5


def test_steady_state_relax_sh_720p_against_oracle(pkg, api, oracle, hip):
    """BASELINE config 4's denoiser (RELAX_DIFFUSE_SPECULAR_SH) at 1280x720, 36 frames: bit-exact on frames 31..36"""
    run_pair(pkg, api, oracle, hip, ["RELAX_DIFFUSE_SPECULAR_SH"], 1280, 720, 128)


def test_steady_state_config3_720p_against_oracle(pkg, api, oracle, hip):
    """BASELINE config 3's denoiser set (REBLUR_DIFFUSE_SPECULAR + SIGMA_SHADOW_TRANSLUCENCY) at 1280x720, 36 frames: bit-exact on frames 31..36,
    the RGBA8 shadow included"""
    run_pair(pkg, api, oracle, hip, ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW_TRANSLUCENCY"], 1280, 720, 128)


def test_steady_state_4k_against_oracle(pkg, api, oracle, hip):
    """The headline AT THE HEADLINE'S SIZE in the regime the headline is timed in (VERDICT r5 "what's weak" 1b): REBLUR_DIFFUSE_SPECULAR 3840x2160 at
    bench.py's settings_of, 34 frames from a restart (the bench pre-rolls 27 + 5 warm-up = 32 before its first timed frame), frames 33 and 34
    compared bit for bit - every OUT_* plane and every pool plane. The oracle takes ~1 s per 4K frame on the box's host threads."""
    ho, hg = run_pair(pkg, api, oracle, hip, ["REBLUR_DIFFUSE_SPECULAR"], 3840, 2160, 256, frames=34, first_checked=32)
    sat = saturated(hg)
    print("share of accumulation-speed codes at the cap after 34 frames at 4K: %.3f" % sat)
    assert sat > 0.5
