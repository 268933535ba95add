"""GPU-box probe (round 5): bench.py's hw_transcendentals leg shows ONE HistoryFix launch of ~22 ms in its first evented step when the hwt
library runs as the third library of the process. Which step, which kernel, and does it follow the library or the position in the process?"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import torch
import bench

pkg = graft.load_package()
api, synth = pkg.api, pkg.synth
from nrd_sample_amd.harness import Harness

w, h = 3840, 2160
dens = [api.Denoiser.REBLUR_DIFFUSE_SPECULAR]
order = sys.argv[1].split(",") if len(sys.argv) > 1 else ["default", "frozen", "hwt"]
for fl in order:
    b = pkg.hip_backend("cuda:0", flavour=None if fl == "default" else fl)
    scene = synth.Scene(w, h, dolly=0.002, device="cuda:0")
    hz = Harness(b, dens, w, h)
    r = bench.SingleRunner(api, hz, scene, dens, 4, bench.settings_of(api, scene, dens))
    for f in range(34):
        r.step(f, reset=(f == 0))
    torch.cuda.synchronize()
    rows = []
    for f in range(34, 34 + 12):
        r.enable_events(True)
        t0 = time.perf_counter()
        r.step(f, reset=False)
        host_ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        evs = r.events[-1]
        rows.append([round(a.elapsed_time(b_), 3) for a, b_ in evs] + [round(host_ms, 2)])
    print(fl, "per evented step: [CT, fused, HF, Blur, PostBlur, TS, host ms of the step]")
    for row in rows:
        print("   ", row)
    del r, hz, scene
    torch.cuda.empty_cache()
