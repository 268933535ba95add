"""bench.py pieces that need no GPU: defaults of the driver contract (N = 1, K / W that finish in minutes), the camera ping-pong,
and the lookup of the committed PMC traffic table behind `roofline.traffic`."""
import sys

import bench


def test_defaults_follow_the_driver_contract(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup, a.workload) == (1, 48, 32, "reblur_ds_4k")
    assert a.event_stride >= 1 and a.steps % a.event_stride == 0  # every kernel gets steps / stride event samples
    w, h, dens = bench.WORKLOADS[a.workload]
    assert (w, h, dens) == (3840, 2160, ["REBLUR_DIFFUSE_SPECULAR"])  # the configuration BASELINE.json quotes its target on


def test_pingpong_camera_path_is_continuous():
    seq = [bench.pingpong(4, f) for f in range(13)]
    assert seq[:7] == [0, 1, 2, 3, 2, 1, 0]
    assert all(abs(a - b) == 1 for a, b in zip(seq, seq[1:]))  # every step has a neighbour frame as its previous camera
    assert [bench.pingpong(1, f) for f in range(3)] == [0, 0, 0]


def test_traffic_lookup_reads_the_newest_committed_counter_table():
    px = 3840 * 2160
    for name, algorithmic_bpp in (("REBLUR::Blur", 50), ("REBLUR::ClassifyTiles", 16)):
        t = bench.measured_traffic("reblur_ds_4k", name)
        assert isinstance(t, int) and 0.8 * algorithmic_bpp * px < t < 2.0 * algorithmic_bpp * px, (name, t)
    assert bench.measured_traffic("reblur_ds_4k", "REBLUR::NoSuchPass") is None
    assert bench.measured_traffic("no_such_workload", "REBLUR::Blur") is None


def test_multi_gpu_default_is_baseline_config_5(monkeypatch, pkg):
    """--gpus N > 1 without --workload: ONE 7680x4320 frame row-tiled into N bands (strong scaling), BASELINE.json configs[4]"""
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    a = bench.parse()
    assert (a.workload, a.scaling) == ("reblur_ds_8k", "strong")
    assert bench.WORKLOADS[a.workload] == (7680, 4320, ["REBLUR_DIFFUSE_SPECULAR"])
    from nrd_sample_amd import tiler
    bands = [tiler.band_layout(4320, 8, r, 80) for r in range(8)]
    assert sum(b["own_rows"] for b in bands) == 4320 and all(528 <= b["own_rows"] <= 544 and b["own0"] % 16 == 0 for b in bands)
    assert all(a["own1"] == b["own0"] for a, b in zip(bands, bands[1:]))


def test_contract_bytes_table():
    assert bench.contract_bpp(["REBLUR_DIFFUSE_SPECULAR"]) == 352.0
    assert bench.contract_bpp(["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW_TRANSLUCENCY"]) == 428.0
    assert bench.contract_bpp(["RELAX_DIFFUSE_SPECULAR"]) is None  # no figure in the contract's table: reported as null, not guessed
