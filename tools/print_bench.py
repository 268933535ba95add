"""One-line summary of bench.py output files (the last JSON line of each): value, ms per step, roofline fractions, per-pass ms, the no-sky leg.
usage: python tools/print_bench.py <bench output or .jsonl> [...]"""
import json
import sys

for path in sys.argv[1:]:
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        print(path, "no JSON line")
        continue
    for line in (lines if path.endswith(".jsonl") else lines[-1:]):
        d = json.loads(line)
        r = d.get("roofline", {})
        fc = d["config"].get("full_coverage", {})
        print(path, d["config"].get("workload", "")[:60], d["value"], d["ms_per_step"], "dominant frac", r.get("frac"), "pipeline", r.get("pipeline_frac_contract"),
              "traffic same_build", (r.get("traffic_source") or {}).get("same_build"), d.get("passes_ms"), "no sky", fc.get("value"))
