"""Per-kernel issue-rate table from the rocprofv3 --pmc passes of tools/pmc_valu_issue.sh.
usage: valu_issue_report.py <dir prefix> <number of passes> <timed launches per kernel>
Derived columns (per kernel, means over the timed launches):
  valu/wave      SQ_INSTS_VALU / SQ_WAVES                      VALU instructions a wave issues
  waves/simd     SQ_WAVES / 1024                               (256 CUs x 4 SIMDs)
  cyc            kernel duration x shader clock (GRBM_GUI_ACTIVE / duration when collected, else 2.4 GHz)
  cyc/valu       cyc / (valu/wave x waves/simd)                shader cycles per VALU wave-instruction per SIMD, if VALU issue were all a SIMD did:
                                                               tools/ubench/valu_rate.hip measures 2.4-2.6 for v_fma/mul/add_f32 and ~4 for conversions,
                                                               compares, v_med3, integer ops, v_fma_mix at 4+ waves per SIMD
  valu busy      SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CU_CYCLES ...) see the column header in the output
"""
import csv, glob, os, sys
from collections import defaultdict

prefix, npass, timed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
acc = defaultdict(lambda: defaultdict(list))
for i in range(1, npass + 1):
    d = "%s%d" % (prefix, i)
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "nrdhip::" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].replace("nrdhip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (i, r["Dispatch_Id"])
            if key not in seen and r["Dispatch_Id"] in dur:
                seen.add(key)
                acc[k]["_ns%d" % i].append(dur[r["Dispatch_Id"]])


def mean_last(v):
    v = v[-timed:]
    return sum(v) / len(v) if v else float("nan")


print("# issue-rate table: means over the last %d launches of each kernel (the timed region of bench.py); counters are chip totals" % timed)
print("# (every counter of a pass was collected in the same run as the kernel durations quoted for that pass; durations under --pmc are a few % longer than untraced)")
names = sorted(acc)
for k in names:
    m = {c: mean_last(v) for c, v in acc[k].items()}
    ns = m.get("_ns1", float("nan"))
    clk = m.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0 / ns if ns == ns else float("nan")  # GRBM_GUI_ACTIVE sums the 8 XCDs
    if not (clk == clk) or clk < 0.5 or clk > 3.5:
        clk = 2.4
    waves = m.get("SQ_WAVES", float("nan"))
    valu = m.get("SQ_INSTS_VALU", float("nan"))
    vpw = valu / waves if waves else float("nan")
    wps = waves / 1024.0
    cyc = ns * clk
    print("\n%s" % k)
    print("  duration %.1f us, shader clock %.2f GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration) -> %.0f k cycles" % (ns / 1e3, clk, cyc / 1e3))
    print("  waves %.0f (%.1f per SIMD), VALU instructions %.1f M = %.0f per wave" % (waves, wps, valu / 1e6, vpw))
    print("  cycles per VALU wave-instruction per SIMD, if the SIMD did nothing else: %.2f   [ubench ceiling: 2.4-2.6 fma/mul/add, ~4 cvt/cmp/med3/int/fma_mix]" % (cyc / (vpw * wps)))
    busy_cu = m.get("SQ_BUSY_CU_CYCLES", float("nan"))
    print("  SQ_BUSY_CU_CYCLES %.4g / (256 CUs x cycles) = %.2f (share of the launch the CUs hold waves)" % (busy_cu, busy_cu / (256 * cyc)) if busy_cu == busy_cu else "")
    wc = m.get("SQ_WAVE_CYCLES", float("nan"))
    for c, label in (("SQ_ACTIVE_INST_VALU", "VALU instruction executing"), ("SQ_WAIT_INST_ANY", "waiting on any instruction issue (s_waitcnt / dependency)"),
                     ("SQ_WAIT_ANY", "waiting (any)"), ("SQ_ACTIVE_INST_VMEM", "VMEM instruction issuing"), ("SQ_INST_CYCLES_VMEM_RD", "VMEM read issue cycles"),
                     ("SQ_ACTIVE_INST_ANY", "any instruction executing"), ("SQ_ACTIVE_INST_SCA", "scalar instruction executing"), ("SQ_ACTIVE_INST_LDS", "LDS instruction executing")):
        if c in m and wc == wc and wc:
            print("  %-24s %12.4g   = %.3f of SQ_WAVE_CYCLES (%s)" % (c, m[c], m[c] / wc, label))
    for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INST_LEVEL_VMEM",
              "SQ_IFETCH", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES"):
        if c in m:
            extra = ""
            if c == "SQC_ICACHE_MISSES" and m.get("SQC_ICACHE_REQ"):
                extra = "   miss rate %.4f" % (m[c] / m["SQC_ICACHE_REQ"])
            if c in ("SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS") and waves:
                extra = "   = %.1f per wave" % (m[c] / waves)
            print("  %-24s %12.4g%s" % (c, m[c], extra))
