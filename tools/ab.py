#!/usr/bin/env python3
"""A/B timing of prebuilt library variants (_variants/<name>.so) on ONE GPU box, alternating variants round by round.
usage: python tools/ab.py --rounds 3 --workload reblur_ds_4k [--full-coverage] [--bench-args "..."] v1 v2 ...
Every run is `python bench.py` with the variant copied over nrd-sample_amd/csrc/libnrdhip.so (restored afterwards); prints one line per
run (Mpixels/s + per-pass ms) and the per-variant medians at the end."""
import argparse, json, os, shutil, statistics, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "nrd-sample_amd", "csrc", "libnrdhip.so")
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--workload", default="reblur_ds_4k")
ap.add_argument("--full-coverage", action="store_true", help="also report the no-sky leg")
ap.add_argument("--bench-args", default="")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
keep = LIB + ".ab_keep"
shutil.copy(LIB, keep)
res = {v: [] for v in a.variants}
try:
    for r in range(a.rounds):
        for v in a.variants:
            shutil.copy(os.path.join(ROOT, "_variants", v + ".so"), LIB)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", a.workload, "--no-cpu-baseline", "--no-frozen-leg", "--no-young-leg", "--no-graph-leg"] + \
                  ([] if a.full_coverage else ["--no-full-coverage"]) + a.bench_args.split()
            try:
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=300).stdout
                d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            except Exception as e:
                print(v, "FAILED", e, flush=True)
                continue
            row = {"value": d["value"], "ms": d["ms_per_step"], "passes": d.get("passes_ms", {})}
            if a.full_coverage and "full_coverage" in d["config"] and "value" in d["config"]["full_coverage"]:
                row["fc_value"], row["fc_ms"] = d["config"]["full_coverage"]["value"], d["config"]["full_coverage"]["ms_per_step"]
                row["fc_passes"] = d["config"]["full_coverage"]["passes_ms"]
            res[v].append(row)
            print("%-32s %8.0f Mpix/s %.4f ms | %s%s" % (v, row["value"], row["ms"], " ".join("%s=%.4f" % (k.split("::")[1][:9], t) for k, t in row["passes"].items()),
                                                       (" | full-coverage %8.0f %.4f ms" % (row["fc_value"], row["fc_ms"])) if "fc_value" in row else ""), flush=True)
finally:
    shutil.copy(keep, LIB)
    os.remove(keep)
print("---- medians (%s, %d rounds)" % (a.workload, a.rounds))
base = None
for v in a.variants:
    if not res[v]:
        continue
    med = statistics.median(x["value"] for x in res[v])
    base = base or med
    names = list(res[v][0]["passes"].keys())
    line = "%-32s %8.0f Mpix/s (%+.1f %%) | %s" % (v, med, (med / base - 1) * 100, " ".join("%s=%.4f" % (k.split("::")[1][:9], statistics.median(x["passes"][k] for x in res[v])) for k in names))
    if "fc_value" in res[v][0]:
        line += " | full-coverage %8.0f" % statistics.median(x["fc_value"] for x in res[v])
    print(line)
