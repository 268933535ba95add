#!/bin/bash
# Runs ON THE GPU BOX: effective shader clock and a few SQ counters per kernel for library variants (_variants/<v>.so):
# rocprofv3 --pmc (own run, kernel-trace only) over a short bench; prints per-kernel mean duration, GRBM_GUI_ACTIVE / duration.
#   usage: tools/clock_probe.sh "<counters>" v1 v2 ...
set -u
PMC=$1; shift
ROOT=$(pwd); export TMPDIR=/tmp
for v in "$@"; do
  cp _variants/$v.so nrd-sample_amd/csrc/libnrdhip.so
  rm -rf /tmp/cp_$v; mkdir -p /tmp/cp_$v
  (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/cp_$v -o p -- python $ROOT/bench.py --workload ${WL:-reblur_ds_4k} --steps 12 --warmup 32 --no-cpu-baseline > /tmp/cp_$v.log 2>&1)
  python - $v /tmp/cp_$v <<'PY'
import csv, glob, sys, os
from collections import defaultdict
v, d = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "nrdhip::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].replace("nrdhip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur: acc[k]["_ns"].append(dur[r["Dispatch_Id"]][1])
for k, cs in sorted(acc.items()):
    last = {c: vals[-12 * (len(vals) // 44 or 1):] for c, vals in cs.items()}
    m = {c: sum(x) / len(x) for c, x in last.items() if x}
    line = "%s %-52s ns=%8.0f" % (v, k[:52], m.get("_ns", 0))
    for c in sorted(m):
        if c != "_ns": line += " %s=%.4g" % (c, m[c])
    if "GRBM_GUI_ACTIVE" in m and m.get("_ns"): line += " clockGHz=%.3f" % (m["GRBM_GUI_ACTIVE"] / m["_ns"])
    print(line)
PY
done
