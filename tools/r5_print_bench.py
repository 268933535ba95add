"""One-line summary of bench.py output files (the last JSON line of each): value, ms per step, roofline fractions, per-pass ms, the no-sky leg.
usage: python tools/r5_print_bench.py <bench output> [...]"""
import json
import sys

for path in sys.argv[1:]:
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        print(path, "no JSON line")
        continue
    d = json.loads(lines[-1])
    r = d["roofline"]
    fc = d["config"].get("full_coverage", {})
    print(path, d["value"], d["ms_per_step"], "dominant frac", r["frac"], "pipeline", r.get("pipeline_frac_contract"),
          "traffic same_build", (r.get("traffic_source") or {}).get("same_build"), d["passes_ms"], "no sky", fc.get("value"))
