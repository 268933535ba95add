"""bench.py's default (N = 1) control flow, end to end, without a GPU: main() runs on the host-emulated kernels with torch.cuda's
streams / events / synchronisation replaced by host stand-ins and the workload shrunk to a few tiles. What is checked is the contract of
the JSON line the driver parses - every key, the roofline and cpu_baseline objects, the extra legs (full coverage, upstream-formulas
flavour, graph replay) - and that nothing in the flow raises; the numbers themselves mean nothing here."""
import contextlib
import json
import sys
import time

import pytest


class FakeEvent:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return max((other.t - self.t) * 1e3, 1e-6)

    def synchronize(self):
        pass


class FakeStream:
    def __init__(self, device=None):
        self.cuda_stream = 0

    def synchronize(self):
        pass


def test_default_bench_line_on_the_emulated_backend(monkeypatch, capsys, pkg, emulated, emulated_upstream):
    import torch

    import bench

    for name, value in (("is_available", lambda: True), ("device_count", lambda: 1), ("set_device", lambda d: None), ("synchronize", lambda *a: None),
                        ("empty_cache", lambda: None), ("Event", FakeEvent), ("Stream", FakeStream), ("stream", lambda s: contextlib.nullcontext()),
                        ("current_stream", lambda *a: FakeStream())):
        monkeypatch.setattr(torch.cuda, name, value)
    monkeypatch.setattr(pkg, "hip_backend", lambda device, flavour=None: emulated_upstream if flavour else emulated)
    real_scene = pkg.synth.Scene
    monkeypatch.setattr(pkg.synth, "Scene", lambda *a, **kw: real_scene(*a, **dict(kw, device="cpu")))
    monkeypatch.setitem(bench.WORKLOADS, "reblur_ds_4k", (96, 64, ["REBLUR_DIFFUSE_SPECULAR"]))
    monkeypatch.setattr(bench, "GRAPH_LEG_BAND", (64, 32))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "8", "--warmup", "2", "--unique-frames", "2"])
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # ONE JSON line
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "passes_ms"):
        assert key in d, key
    assert (d["n_gpus"], d["steps"], d["warmup"], d["unit"], d["higher_is_better"], d["scaling"], d["vs_baseline"], d["data"]) == (1, 8, 2, "Mpixels/s", True, "weak", None, "synthetic")
    assert d["value"] > 0 and abs(d["value"] - 96 * 64 * 8 / (d["ms_per_step"] * 8e-3) / 1e6) <= 0.0051  # (value is rounded to 2 decimals)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes_contract"] == 352.0 and r["pipeline_frac_contract"] >= 0 and "traffic" in r  # (rounded to 4 decimals: 0 at emulation speed)
    assert list(d["passes_ms"]) == ["REBLUR::ClassifyTiles", "REBLUR::PrePass", "REBLUR::TemporalAccumulation", "REBLUR::HistoryFix", "REBLUR::Blur",
                                    "REBLUR::PostBlur", "REBLUR::TemporalStabilization"]
    c = d["config"]
    assert "workload" in c and 0.0 <= c["sky_fraction"] <= 1.0
    assert c["full_coverage"]["sky_fraction"] == 0.0 and c["full_coverage"]["value"] > 0
    assert c["upstream_formulas"]["value"] > 0
    assert set(c["graph_replay"]) == {"workload", "band_64x32"} and c["graph_replay"]["workload"]["graph_stats"]["direct"] > 0  # (no graphs in the emulation)
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["value"] > 0 and b["cores"] >= 1 and "sample" in b and b["unit"] == "Mpixels/s"
