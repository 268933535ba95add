"""TEST INFRASTRUCTURE (tests/test_bench_flow.py): runs bench.main() without a GPU - torch.cuda's streams / events / synchronisation
replaced by host stand-ins, the HIP backend by the host-emulated kernels, the workloads shrunk to a few tiles - in this process (imported
by the N = 1 test) or as the per-rank script of a torch.distributed.run launch (the N = 2 test: NRD_BENCH_DRYRUN_BACKEND=gloo,
NRD_BENCH_DEVICE=cpu)."""
import contextlib
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


class FakeEvent:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return max((other.t - self.t) * 1e3, 1e-6)

    def synchronize(self):
        pass


class FakeStream:
    def __init__(self, device=None):
        self.cuda_stream = 0

    def synchronize(self):
        pass


CUDA_STANDINS = (("is_available", lambda: True), ("device_count", lambda: 1), ("set_device", lambda d: None), ("synchronize", lambda *a: None),
                 ("empty_cache", lambda: None), ("Event", FakeEvent), ("Stream", FakeStream), ("stream", lambda s: contextlib.nullcontext()),
                 ("current_stream", lambda *a: FakeStream()))


def patch_for_cpu(setattr_, pkg, bench, emulated, emulated_frozen, workloads, graph_band):
    """setattr_(obj, name, value): monkeypatch.setattr in a test, plain setattr in a worker process"""
    import torch

    for name, value in CUDA_STANDINS:
        setattr_(torch.cuda, name, value)
    setattr_(pkg, "hip_backend", lambda device, flavour=None: emulated_frozen if flavour == "frozen" else emulated)  # (the hwt leg runs on the default emulation: control flow only)
    real_scene = pkg.synth.Scene
    setattr_(pkg.synth, "Scene", lambda *a, **kw: real_scene(*a, **dict(kw, device="cpu")))
    setattr_(bench, "WORKLOADS", dict(bench.WORKLOADS, **workloads))
    setattr_(bench, "GRAPH_LEG_BAND", graph_band)
    setattr_(bench, "FLAVOUR_DISTANCE", (64, 48, 3))


if __name__ == "__main__":
    import __graft_entry__ as graft

    pkg = graft.load_package()
    import bench

    emu = pkg.api.Backend(graft.build_emulated(), "nrdhip_", "cpu")
    ww, hh = (int(v) for v in os.environ.get("NRD_TEST_FRAME", "48,448").split(","))
    patch_for_cpu(setattr, pkg, bench, emu, emu, {"reblur_ds_8k": (ww, hh, ["REBLUR_DIFFUSE_SPECULAR"])}, (64, 32))
    bench.main()
