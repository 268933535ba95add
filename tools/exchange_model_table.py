#!/usr/bin/env python3
"""the predicted cost of the row-tiled 8K frame (BASELINE config 5) at N = 2, 4, 8 from the exchange plan (bytes per dispatch
and neighbour) and 1-GPU pass times of the 8K frame scaled to a band (tiler.exchange_overlap_model). Runs on the CPU (the oracle backend
provides the dispatch list). usage: python tools/exchange_model_table.py [bench_workloads.jsonl with a reblur_ds_8k line]"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

pkg = g.load_package()
api = pkg.api
from nrd_sample_amd import tiler

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_bench_workloads.jsonl")
ms8k = None
for l in open(src):
    d = json.loads(l)
    if "reblur_ds_8k" in d["config"]["workload"]:
        ms8k, one_gpu = d["passes_ms"], d["ms_per_step"]
assert ms8k, "no reblur_ds_8k line in " + src
den = api.Denoiser.REBLUR_DIFFUSE_SPECULAR
orc = g.oracle_backend()
W, H = 7680, 4320
print("1 GPU, 8K: %.3f ms per frame (%s)" % (one_gpu, os.path.basename(src)))
print("| N | band rows | link GB/s, latency us | compute ms | unhidden exchange ms | predicted ms / frame | efficiency vs 1 GPU / N | serial (nothing hidden) ms |")
print("|---|---|---|---|---|---|---|---|")
for n in (2, 4, 8):
    rows = H // n
    band = tiler.BandHarness(orc, [den], 128, 1280, 1 if n > 2 else 0, max(n, 2))  # the plan depends on neighbours and reach only, not on the frame size
    t = tiler.Tiler(band, None)
    disp = band.nrd.dispatches([int(den)])
    planes = {(pool << 16) | i: p["bpt"] for pool in (0, 1) for i, p in enumerate(band.nrd.pools[pool])}
    pb = tiler.plan_bytes(t._plan([int(den)], disp), planes, W)
    halo = band.halo
    # a band computes its own rows; the strips are part of them (halo rows are computed by ClassifyTiles only)
    pm = [ms8k[d["name"]] * rows / H for d in disp]
    for gbs, lat in ((50, 20), (75, 10), (35, 50)):
        m = tiler.exchange_overlap_model([d["name"] for d in disp], pm, pb, rows, 1 if n == 2 else 2, gbs, lat)
        print("| %d | %d | %d, %d | %.3f | %.3f | %.3f | %.2f | %.3f |" % (n, rows, gbs, lat, m["compute_ms"], m["unhidden_exchange_ms"], m["predicted_frame_ms"],
                                                                  (one_gpu / n) / m["predicted_frame_ms"], m["serial_frame_ms"]))
    if n == 8:
        worst = tiler.exchange_overlap_model([d["name"] for d in disp], pm, pb, rows, 2, 50, 20)
        for r in worst["per_dispatch"]:
            print("    %-40s compute %.3f interior %.3f exchange %.3f unhidden %.3f (%d bytes strips-first)" % (r["dispatch"], r["compute_ms"], r["interior_ms"], r["exchange_ms"], r["unhidden_ms"], r["strips_first_bytes"]))
