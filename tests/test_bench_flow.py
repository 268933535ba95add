"""bench.py's default (N = 1) control flow, end to end, without a GPU: main() runs on the host-emulated kernels with torch.cuda's
streams / events / synchronisation replaced by host stand-ins and the workload shrunk to a few tiles. What is checked is the contract of
the JSON line the driver parses - every key, the roofline and cpu_baseline objects, the extra legs (full coverage, frozen-formulas
flavour, graph replay) - and that nothing in the flow raises; the numbers themselves mean nothing here."""
import json
import os
import subprocess
import sys

import pytest


import bench_worker


def test_default_bench_line_on_the_emulated_backend(monkeypatch, capsys, pkg, emulated, emulated_frozen):
    import torch

    import bench

    bench_worker.patch_for_cpu(monkeypatch.setattr, pkg, bench, emulated, emulated_frozen, {"reblur_ds_4k": (96, 64, ["REBLUR_DIFFUSE_SPECULAR"])}, (64, 32))
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
    assert list(d["passes_ms"]) == ["REBLUR::ClassifyTiles", "REBLUR::PrePassTemporalAccumulation", "REBLUR::HistoryFix", "REBLUR::Blur",
                                    "REBLUR::PostBlur", "REBLUR::TemporalStabilization"]
    c = d["config"]
    assert "workload" in c and 0.0 <= c["sky_fraction"] <= 1.0
    assert c["full_coverage"]["sky_fraction"] == 0.0 and c["full_coverage"]["value"] > 0
    assert c["frozen_formulas"]["value"] > 0
    dist_ = c["frozen_formulas"]["distance_from_default"]  # PSNR and the share of values > 1 fp16 ULP apart, per output, default vs frozen flavour
    assert dist_["frames"] == 3 and all(0.0 <= dist_[k]["ulp_gt1_frac"] <= 1.0 and dist_[k]["psnr_db"] > 20.0 for k in ("out_diff", "out_spec"))
    assert c["hw_transcendentals"]["value"] > 0 and "distance_from_default" in c["hw_transcendentals"]  # the optional flavour's leg
    assert c["without_preroll"]["value"] > 0 and "warm-up" in c["without_preroll"]["what"]  # both timing regimes in one line
    ts = r["traffic_source"]  # a table lookup is labelled as one: which table, of which library, and whether that is the library timed here
    assert ts is None or (ts["table"].startswith("profiles/") and "same_build" in ts and ts["timed_library_sha256_12"])
    assert set(c["graph_replay"]) == {"workload", "band_64x32"} and c["graph_replay"]["workload"]["graph_stats"]["direct"] > 0  # (no graphs in the emulation)
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["value"] > 0 and b["cores"] >= 1 and "sample" in b and b["unit"] == "Mpixels/s"
    ss = b["same_size"]  # one thread and all threads on the SAME frame
    assert ss["one_thread"] > 0 and ss["all_threads"] > 0 and ss["threads"] >= 1 and ss["ratio"] > 0


def run_two_ranks(tmp_path, extra, world=2, frame="48,448"):
    port = 29600 + os.getpid() % 300 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_worker.py"), "--gpus", str(world), "--steps", "8", "--warmup", "2", "--unique-frames", "2",
           "--no-cpu-baseline"] + extra
    env = dict(os.environ, NRD_BENCH_DRYRUN_BACKEND="gloo", NRD_BENCH_DEVICE="cpu", OMP_NUM_THREADS="2", NRD_TEST_FRAME=frame)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_bench_line_on_the_emulated_backend(emulated, tmp_path):
    """the N > 1 flow (BASELINE config 5's mode: ONE frame row-tiled into bands, strong scaling) with two gloo ranks on the emulated
    kernels: the line of record, per-rank GPU-busy times, the balanced bands, the C++ tiler leg and the bit-identity check"""
    d = run_two_ranks(tmp_path, [])
    c = d["config"]
    assert (d["n_gpus"], d["scaling"], d["steps"]) == (2, "strong", 8) and d["value"] > 0
    assert sum(c["band_rows"]) == 448 and len(c["rank_ms"]) == 2 and all(t > 0 for t in c["rank_ms"])
    assert c["tiled_bit_identical"] is True, c.get("tiled_bit_identical_detail")
    # round 6: --tiler auto tries the C++ tiler FIRST (two-frame probe on every rank) and the value of record comes from it
    assert c["native_tiler"]["used_for_value"] is True and c["native_tiler"]["fallback_reason"] is None, c["native_tiler"]
    assert "native tiler" in c["workload"]
    assert "roofline" in d and d["roofline"]["bound"] == "hbm"


def test_two_rank_bench_python_tiler_with_native_leg(emulated, tmp_path):
    """--tiler python: the Python tiler is the value of record and the C++ tiler runs as a second leg beside it (rounds 3-5's default)"""
    d = run_two_ranks(tmp_path, ["--tiler", "python"])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and all(t > 0 for t in c["rank_ms"])
    assert "value" in c["native_tiler"], c["native_tiler"]
    assert c["tiled_bit_identical"] is True, c.get("tiled_bit_identical_detail")


def test_two_rank_bench_falls_back_to_the_python_tiler(emulated, tmp_path, monkeypatch):
    """--tiler auto, the C++ tiler raising on every rank: the ranks agree through one all_reduce, move to the Python tiler and the line says why"""
    monkeypatch.setenv("NRD_BENCH_NATIVE_FAIL_RANK", "all")
    d = run_two_ranks(tmp_path, ["--no-identity-check"])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and all(t > 0 for t in c["rank_ms"])
    assert c["native_tiler"]["used_for_value"] is False and "injected failure" in c["native_tiler"]["fallback_reason"], c["native_tiler"]
    assert "python tiler" in c["workload"]


def test_two_rank_bench_restarts_with_the_python_tiler_when_the_native_probe_hangs(emulated, tmp_path, monkeypatch):
    """--tiler auto, the C++ tiler raising on ONE rank only: the other rank's probe then waits for rows that never come (what a stuck ncclRecv
    looks like). After --native-deadline every rank replaces itself with the same command + --tiler python, the restarted ranks meet again
    behind a key prefix of the launcher's store, and the line of record carries the reason"""
    monkeypatch.setenv("NRD_BENCH_NATIVE_FAIL_RANK", "1")
    d = run_two_ranks(tmp_path, ["--no-identity-check", "--native-deadline", "20"])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and all(t > 0 for t in c["rank_ms"])
    assert c["native_tiler"]["used_for_value"] is False and "not finished after 20 s" in c["native_tiler"]["fallback_reason"], c["native_tiler"]
    assert "python tiler" in c["workload"]


def test_two_rank_bench_watchdog_prints_the_line_and_ends_the_run(emulated, tmp_path):
    """--extras-deadline 0: the watchdog fires while the extras run - rank 0 must still print the complete line of record, with the
    unfinished extras marked, and every rank must leave with exit code 0"""
    d = run_two_ranks(tmp_path, ["--extras-deadline", "0"])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(c["rank_ms"]) == 2
    assert c["tiled_bit_identical"] is None and "watchdog" in c["tiled_bit_identical_detail"]
    assert "error" in c["native_tiler"] or "value" in c["native_tiler"] or "used_for_value" in c["native_tiler"]


def test_four_rank_bench_line_on_the_emulated_backend(emulated, tmp_path):
    """four bands: interior ranks exchange rows with two neighbours; the line carries four per-rank times and four band heights"""
    d = run_two_ranks(tmp_path, ["--no-native-leg"], world=4, frame="32,1280")
    c = d["config"]
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and len(c["rank_ms"]) == 4 and len(c["band_rows"]) == 4 and sum(c["band_rows"]) == 1280
    assert all(r % 16 == 0 for r in c["band_rows"]) and c["tiled_bit_identical"] is True, c


def test_eight_rank_bench_line_on_the_emulated_backend(emulated, tmp_path):
    """the driver's largest launch shape (N = 8, one rank per GPU there; eight gloo ranks on the host here), native tiler leg included"""
    d = run_two_ranks(tmp_path, [], world=8, frame="32,2560")
    c = d["config"]
    assert d["n_gpus"] == 8 and len(c["rank_ms"]) == 8 and len(c["band_rows"]) == 8 and sum(c["band_rows"]) == 2560
    assert c["tiled_bit_identical"] is True and c["native_tiler"]["used_for_value"] is True, c
