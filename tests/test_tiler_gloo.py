"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the row tiler with the oracle as compute backend must be bit-identical
to the single-instance run on the whole frame ("N-GPU output must be bit-identical to 1-GPU")."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import util

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def apply_mode(api, reblur_settings, mode):
    """"cb": the sample's default operating point - checkerboarded inputs + hit distance reconstruction - plus anti-firefly"""
    if mode == "cb":
        reblur_settings.checkerboardMode = int(api.CheckerboardMode.WHITE)
        reblur_settings.hitDistanceReconstructionMode = int(api.HitDistanceReconstructionMode.AREA_5X5)
        reblur_settings.enableAntiFirefly = True


def mode_frame_hook(pkg, mode, f, fr):
    if mode != "cb":
        return
    rng = np.random.default_rng(100 + f)
    for key in ("diff", "spec"):
        a = np.array(fr[key])
        a[rng.random(a.shape[:2]) < 0.4, 3] = 0
        fr[key] = a
    fr.update(pkg.harness.to_checkerboard(fr, f, white=True))


@pytest.mark.parametrize("world,frame_h,mode", [(2, 224, "default"), (3, 336, "default"), (2, 224, "cb"),
                                                 (2, 640, "default"), (3, 1008, "cb")])  # tall bands: boundary strips first, exchange overlapped
def test_row_tiling_bit_identical(tmp_path, pkg, api, oracle, world, frame_h, mode):
    run_tiled_and_compare(tmp_path, pkg, api, oracle, world, frame_h, mode, "oracle")


def test_exchange_plan_defers_rows_only_the_next_frame_reads(pkg, api, oracle):
    """Tiler._plan: strips-first rows ("now") are bounded by the reach of the later dispatches of the same frame; every permanent
    plane that survives the frame travels with the full halo in the deferred list ("later"), transient planes never do"""
    from nrd_sample_amd import tiler

    D = api.Denoiser
    for den in (D.REBLUR_DIFFUSE_SPECULAR, D.RELAX_DIFFUSE_SPECULAR_SH, D.SIGMA_SHADOW_TRANSLUCENCY):
        band = tiler.BandHarness(oracle, [den], 128, 1280, 1, 4)
        t = tiler.Tiler(band, None)
        disp = band.nrd.dispatches([int(den)])
        plan = t._plan([int(den)], disp)
        assert len(plan) == len(disp)
        deferred = set()
        for i, (now, later) in enumerate(plan):
            reach = max([d["halo_rows"] for d in disp[i + 1:]] + [0])
            assert all(0 < rows <= min(reach, band.halo) for _, rows in now), (disp[i]["name"], now, reach)
            for code, rows, skip in later:
                assert code >> 16 == 0 and rows == band.halo and code in disp[i]["written"]
                assert skip == dict(now).get(code, 0) < rows  # the rows nearest the edge are not sent twice
                deferred.add(code)
        assert plan[-1][0] == []  # nothing after the last dispatch reads its outputs this frame: no strips, no wait
        written_last = {}  # permanent plane -> last dispatch that writes it
        for i, d in enumerate(disp):
            for c in d["written"]:
                if c >> 16 == 0:
                    written_last[c] = i
        for c, i in written_last.items():  # its final version reaches the neighbours with the full halo, deferred or at once
            assert any(e[0] == c and e[1] == band.halo for e in plan[i][1] + plan[i][0]), (disp[i]["name"], c)
        assert deferred <= set(written_last)


def test_row_tiling_emulated_kernels_bit_identical(tmp_path, pkg, api, oracle, emulated):
    """the kernel sources themselves (compiled for the host) on two bands: rows stored at a band offset, nrdhip_denoise_rows strips,
    halo rows owned by the neighbour - against the single-instance oracle run"""
    run_tiled_and_compare(tmp_path, pkg, api, oracle, 2, 448, "default", "emu")


@pytest.mark.gpu
@pytest.mark.parametrize("world,frame_h,mode", [(2, 640, "default"), (2, 704, "cb")])
def test_row_tiling_hip_bit_identical(tmp_path, pkg, api, oracle, hip, world, frame_h, mode):
    """the real kernels, two ranks sharing the one GPU of the box (gloo moves the rows): exercises nrdhip_denoise_rows, the strip /
    interior split and the stream ordering of the overlapped exchange; the result must equal the single-instance oracle run"""
    run_tiled_and_compare(tmp_path, pkg, api, oracle, world, frame_h, mode, "hip")


def run_tiled_and_compare(tmp_path, pkg, api, oracle, world, frame_h, mode, backend):
    w, nframes, halo = 96, 3, 80
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(HERE, "tiler_worker.py"), str(tmp_path), str(w), str(frame_h), str(nframes), str(halo), mode, backend]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    # single-instance run over the whole frame
    D = api.Denoiser
    dens = [D.REBLUR_DIFFUSE_SPECULAR, D.SIGMA_SHADOW_TRANSLUCENCY]
    scene = pkg.synth.Scene(w, frame_h, dolly=0.03)
    st = {D.REBLUR_DIFFUSE_SPECULAR: api.ReblurSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1),
          D.SIGMA_SHADOW_TRANSLUCENCY: api.SigmaSettings(lightDirection=list(scene.sun))}
    apply_mode(api, st[D.REBLUR_DIFFUSE_SPECULAR], mode)
    keep = []
    hz = util.run_frames(api, pkg.harness, oracle, scene, dens, nframes, settings=st, keep=keep, frame_hook=lambda f, fr: mode_frame_hook(pkg, mode, f, fr))
    parts = [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]
    for f in range(nframes):
        for key in ("out_diff", "out_spec", "out_shadow"):
            tiled = np.concatenate([p["f%d_%s" % (f, key)] for p in parts], 0)
            assert np.array_equal(tiled, keep[f][key]), (f, key)
    hist = np.concatenate([p["history"] for p in parts], 0)
    assert np.array_equal(hist, hz.pool("REBLUR::History"))
    assert sum(int(p["bytes"][0]) for p in parts) > 0
    if frame_h // world >= 320:  # tall bands take the overlapped path: boundary strips, exchange in flight, interior
        assert all(int(p["split"][0]) >= 3 * 7 for p in parts)  # every REBLUR dispatch of every frame at least
    assert [int(p["own0"][0]) for p in parts] == sorted(int(p["own0"][0]) for p in parts)
