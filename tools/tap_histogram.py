"""Runs ON THE GPU BOX with a library built with -DNRD_DEBUG_COUNTERS (tools/build_variant.sh hist "-DNRD_DEBUG_COUNTERS" tools/variants/lab_diag_hist_pair.patch):
distribution of tap distances (Chebyshev, pixels) in the three spatial passes of the default bench workload, over all frames
(warm-up included) and over the steady state only."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import bench
import torch
pkg = graft.load_package()
api, synth = pkg.api, pkg.synth
from nrd_sample_amd.harness import Harness
wl = sys.argv[1] if len(sys.argv) > 1 else "reblur_ds_4k"
w, h, names = bench.WORKLOADS[wl]
dens = [api.Denoiser[n] for n in names]
hip = pkg.hip_backend("cuda:0")
scene = synth.Scene(w, h, dolly=0.002, device="cuda:0")
hz = Harness(hip, dens, w, h)
runner = bench.SingleRunner(api, hz, scene, dens, 4, bench.settings_of(api, scene, dens))
def read():
    out = (C.c_ulonglong * 24)()
    tot = [0] * 24
    for fn in ("nrdhip_debug_counters_p0", "nrdhip_debug_counters_p1"):
        getattr(hip.lib, fn)(out)
        tot = [a + b for a, b in zip(tot, out)]
    return tot
prev = read()
for phase, frames in (("warm-up frames 0..31", range(0, 32)), ("steady frames 32..47", range(32, 48))):
    for f in frames:
        runner.step(f, reset=(f == 0))
    torch.cuda.synchronize()
    cur = read()
    d = [c - p for c, p in zip(cur, prev)]
    prev = cur
    print(phase)
    for v, name in enumerate(("PrePass", "Blur", "PostBlur")):
        row = d[v * 8:v * 8 + 8]
        n = max(sum(row), 1)
        cum, acc = [], 0
        for x in row:
            acc += x
            cum.append(100.0 * acc / n)
        print("  %-8s taps=%.3g  cumulative %% within <=2,4,8,12,16,24,32,inf px: %s" % (name, n, " ".join("%.1f" % c for c in cum)))
