"""Summarises the rocprofv3 outputs written by tools/profile_gpu.sh into small per-kernel tables
(gpurun_out/profiles/<tag>_kernel_stats_<workload>.csv, <tag>_pmc_<workload>.csv)."""
import csv, glob, json, os, sys
from collections import defaultdict

tag, wl, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.getcwd()
out = os.path.join(root, "gpurun_out", "profiles")
os.makedirs(out, exist_ok=True)
W, H = {"4k": (3840, 2160), "1440p": (2560, 1440), "1080p": (1920, 1080)}[wl.rsplit("_", 1)[1]]
px = W * H


def short(name):
    n = name.replace("nrdhip::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


# kernel stats (rocprofv3 --stats)
for f in glob.glob(os.path.join(root, "gpurun_out", "prof_" + tag, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, "%s_kernel_stats_%s.csv" % (tag, wl)), "w") as o:
        o.write("kernel,calls,total_ns,avg_ns,percent\n")
        for r in rows:
            if "nrdhip::" in r["Name"]:
                o.write('"%s",%s,%s,%.0f,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))

# steady-state view from the per-dispatch trace: mean over the LAST `steps` launches of each kernel (the timed region of the
# bench; rocprofv3's own --stats average above also spans the warm-up launches, where histories are short and the passes slower)
for f in glob.glob(os.path.join(root, "gpurun_out", "prof_" + tag, "**", "*kernel_trace.csv"), recursive=True):
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "nrdhip::" in r["Kernel_Name"]:
            per[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(out, "%s_kernel_steady_%s.csv" % (tag, wl)), "w") as o:
        o.write("kernel,launches_total,timed_launches,avg_ns_timed_region,avg_ns_all\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1][-steps:])):
            t = v[-steps:]
            o.write('"%s",%d,%d,%.0f,%.0f\n' % (k, len(v), len(t), sum(t) / len(t), sum(v) / len(v)))

# PMC passes: per kernel, per counter: mean over launches
acc = defaultdict(lambda: defaultdict(list))
for d in glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_*" % tag)):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "nrdhip::" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
with open(os.path.join(out, "%s_pmc_%s.csv" % (tag, wl)), "w") as o:
    o.write("kernel,launches," + ",".join(counters) + ",fetch_B_per_px(x2 gfx950 correction),write_B_per_px\n")
    summary = {}
    for k, cs in sorted(acc.items()):
        n = max(len(v) for v in cs.values())
        mean = {c: (sum(cs[c]) / len(cs[c]) if c in cs else float("nan")) for c in counters}
        # FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; FETCH_SIZE counts 128-B requests as 64 B on gfx950 (guide, HBM section)
        fetch = mean.get("FETCH_SIZE", float("nan")) * 1024 * 2 / px
        write = mean.get("WRITE_SIZE", float("nan")) * 1024 / px
        o.write('"%s",%d,%s,%.1f,%.1f\n' % (k, n, ",".join("%.0f" % mean[c] for c in counters), fetch, write))
        summary[k] = {"fetch_bytes_per_launch": mean.get("FETCH_SIZE", 0) * 2048, "write_bytes_per_launch": mean.get("WRITE_SIZE", 0) * 1024}
# which build the counters were collected on: bench.py labels roofline.traffic with it and says whether the library it TIMES is the same one
import hashlib
lib = os.path.join(root, "nrd-sample_amd", "csrc", "libnrdhip.so")
summary["_measured_on"] = {"library_sha256_12": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:12] if os.path.exists(lib) else None, "tag": tag, "workload": wl}
json.dump(summary, open(os.path.join(out, "%s_hbm_traffic_%s.json" % (tag, wl)), "w"), indent=1)
print(open(os.path.join(out, "%s_pmc_%s.csv" % (tag, wl))).read())
