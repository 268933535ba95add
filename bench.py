#!/usr/bin/env python3
"""bench.py - Mpixels/s of the full REBLUR diffuse+specular pipeline on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload reblur_ds_4k|...]

A "step" is one frame through every pass of the denoiser (ClassifyTiles, PrePass, TemporalAccumulation, HistoryFix, Blur,
PostBlur, TemporalStabilization) with all inputs already resident in HBM. N = 1: the 3840x2160 frame BASELINE.json quotes its
target on. N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE.json config 5 - ONE 7680x4320 frame
(--workload reblur_ds_8k, the default for N > 1) row-tiled into N bands, strong scaling; every rank exchanges halo rows with
its <= 2 row neighbours between passes (SURVEY.md 8e scheme A) through the C++ row tiler below the C-ABI (ncclSend / ncclRecv
groups on a side stream, --tiler native) or through torch.distributed (backend nccl = RCCL over xGMI, --tiler python). The
default, --tiler auto, takes the value from the C++ tiler after a two-frame probe on every rank and falls back to the Python tiler
with the reason in config.native_tiler. --scaling weak keeps the round-1 mode: every rank owns a full band of the workload's
height (a W x (H N) frame).

One JSON line on stdout (rank 0). `roofline` is computed for the slowest kernel from HIP events recorded on the launch
stream around every dispatch of every 8th step of the timed region (--event-stride; the 14 event records idle the GPU for
~90 us of such a frame - measured r3: 0.998 ms per frame without any, 1.021 at stride 4 - so the remaining steps enqueue the
frame exactly as the sample would, with one Denoise call); `roofline.frac_geometry` / `pipeline_frac_geometry` count the bytes of
pixels with geometry only, and `config.full_coverage` (the same scene without sky) is the leg to hold against the 70 % target;
`cpu_baseline` times the CPU oracle (a scalar C++ port, oracle/, persistent workers) on a bounded sample of the same workload on this
box's host cores, with one thread and all threads on the same 1080p frame beside it - a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL / tensor sharing across ranks fails with the legacy mode)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import __graft_entry__ as graft  # noqa: E402

FLAVOUR_DISTANCE = (1920, 1080, 34)  # width, height, frames of the default-vs-frozen output comparison (config.frozen_formulas.distance_from_default)
GRAPH_LEG_BAND = (7680, 544)  # the band one rank of 8 holds of the 7680x4320 frame (BASELINE config 5): second size of the graph-replay leg
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (width, height, denoisers, synthetic-scene kind)
    "reblur_ds_4k": (3840, 2160, ["REBLUR_DIFFUSE_SPECULAR"]),
    "reblur_d_1080p": (1920, 1080, ["REBLUR_DIFFUSE"]),
    "reblur_ds_sigma_1440p": (2560, 1440, ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW_TRANSLUCENCY"]),
    "reblur_ds_1080p": (1920, 1080, ["REBLUR_DIFFUSE_SPECULAR"]),
    "relax_ds_4k": (3840, 2160, ["RELAX_DIFFUSE_SPECULAR"]),
    "relax_ds_sh_4k": (3840, 2160, ["RELAX_DIFFUSE_SPECULAR_SH"]),  # BASELINE.json configs[3]
    "reblur_ds_sh_4k": (3840, 2160, ["REBLUR_DIFFUSE_SPECULAR_SH"]),
    "reblur_ds_8k": (7680, 4320, ["REBLUR_DIFFUSE_SPECULAR"]),  # BASELINE.json configs[4] on ONE GPU (the 8-GPU run row-tiles 4K bands)
    "sample_passes_4k": (3840, 2160, []),  # the sample's own passes around the denoiser, timed kernel by kernel
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: reblur_ds_4k on one GPU (the headline metric), reblur_ds_8k row-tiled for --gpus N > 1 (BASELINE config 5)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the workload's frame split into N row bands; weak = every rank a full band of a W x (H N) frame")
    ap.add_argument("--tiler", default="auto", choices=["auto", "python", "native"],
                    help="N > 1: native = the C++ row tiler below the C-ABI (nrdhip_tiler_*, RCCL send / recv groups on a side stream: the path "
                         "north_star describes); python = nrd-sample_amd/tiler.py over torch.distributed P2P; auto (default) = native FIRST - a two-frame "
                         "probe on every rank - and the Python tiler with the reason in config.native_tiler when it raises or does not come back")
    ap.add_argument("--native-deadline", type=float, default=180.0, help="--tiler auto: seconds the native tiler's set-up + probe may take before the "
                    "run restarts itself with --tiler python (a hung RCCL call cannot be cancelled inside the process)")
    ap.add_argument("--native-note", default=None, help=argparse.SUPPRESS)  # set by that restart: why the native tiler was abandoned
    ap.add_argument("--no-native-leg", action="store_true", help="N > 1 with the Python tiler: do not run the C++ / RCCL tiler afterwards")
    ap.add_argument("--no-identity-check", action="store_true", help="N > 1, strong scaling: skip the tiled-vs-single-instance comparison")
    ap.add_argument("--even-bands", action="store_true", help="N > 1, strong scaling: split the frame into bands of equal HEIGHT instead of "
                    "equal estimated cost (sky tiles are cheap; the default balances tiles-with-geometry + 0.15 x tiles-without)")
    ap.add_argument("--motion-rows", type=int, default=8, help="row tiling: vertical motion (rows) the stored halo must cover beyond the passes' reach")
    ap.add_argument("--unique-frames", type=int, default=4, help="distinct noisy input frames cycled through (resident in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate-passes", action="store_true", help="1-GPU runs: one dispatch per pass (NRDHIP_FLAG_SEPARATE_PASSES) instead of the "
                    "fused REBLUR::PrePassTemporalAccumulation dispatch")
    ap.add_argument("--no-preroll", action="store_true", help="do not run the untimed pre-roll frames that bring the accumulation to its steady state "
                    "when --warmup is shorter than the accumulation length (the timed region then sees the wider blur radii of young histories)")
    ap.add_argument("--extras-deadline", type=int, default=240,
                    help="N > 1: seconds the native-tiler leg and the bit-identity check may take together before rank 0 prints the line of "
                         "record without them and the run ends (a hang in never-executed transport code must not cost the line)")
    ap.add_argument("--no-graph-leg", action="store_true", help="skip the HIP-graph replay leg (NRDHIP_FLAG_GRAPH), 1-GPU runs")
    ap.add_argument("--no-frozen-leg", action="store_true", help="skip the timed legs on the other build flavours - libnrdhip_frozen.so (the cheaper formulas of "
                    "rounds 1-3) and libnrdhip_hwt.so (hardware transcendentals in the weight arithmetic) - and the distance of their outputs from the default's, 1-GPU runs")
    ap.add_argument("--no-young-leg", action="store_true", help="skip the leg timed WITHOUT the pre-roll (young histories, wider blurs: the regime rounds 1-3 timed), 1-GPU runs")
    ap.add_argument("--no-full-coverage", action="store_true", help="skip the second timed leg (the same scene without sky), 1-GPU runs")
    ap.add_argument("--force-tiled", action="store_true", help="run the row-tiled path even with one rank (exercises the tiler)")
    ap.add_argument("--dolly", type=float, default=0.002, help="camera translation per frame (scene units)")
    ap.add_argument("--preset", default="", help="FILE:INDEX - one of the sample's recorded test presets (tests/golden/sample_tests/*.bin, "
                    "Source/NRDSample.cpp:1787-1901) as operating point of a 1-GPU run: field of view, sun, hit-distance scale, accumulation "
                    "lengths as Sample::PrepareFrame derives them, view direction (the G-buffer stays procedural)")
    ap.add_argument("--roll", type=float, default=0.0, help="camera roll in degrees (1-GPU runs): 90 puts the sky at one SIDE of the frame - "
                    "a layout probe for the XCD tile traversal, not the headline scene")
    ap.add_argument("--checkerboard", action="store_true",
                    help="the sample's default operating point (tracingMode RESOLUTION_HALF): half-width checkerboarded inputs, "
                         "CheckerboardMode::WHITE -> the PrepareInputs pass runs (single-GPU runner)")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="record the per-dispatch HIP events on every S-th timed step (1 = every step); the other steps enqueue the frame "
                         "with one Denoise call. 14 event records per frame cost ~90 us of GPU idle time between the seven kernels "
                         "(measured r3 at 4K: 0.998 ms per frame with no events, 1.021 ms at stride 4), which is instrumentation, not pipeline")
    ap.add_argument("--atrous", type=int, default=0, help="RELAX: atrousIterationNum override (2..8; BASELINE config 4 also asks for an 8-iteration stress run)")
    a = ap.parse_args()
    if a.workload is None:
        a.workload = "reblur_ds_4k" if a.gpus == 1 else "reblur_ds_8k"
    return a


def cpu_run(pkg, orc, denoiser_names, settings_of, device, w, h, warm, frames, threads):
    """Mpixels/s of the CPU oracle on `frames` frames of a w x h instance of the workload after `warm` warm-up frames"""
    api, synth, harness = pkg.api, pkg.synth, pkg.harness
    scene = synth.Scene(w, h, dolly=0.004, device=device, denoiser="RELAX" if denoiser_names[0].startswith("RELAX") else "REBLUR")
    dens = [api.Denoiser[n] for n in denoiser_names]
    hz = harness.Harness(orc, dens, w, h)
    orc.lib.orc_set_threads(hz.nrd.handle, threads)
    st = settings_of(api, scene, dens)
    data = []
    for f in range(warm + frames):
        fr = scene.frame(f)
        data.append({k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in fr.items()})
    planes = [hz.upload(d) for d in data]
    for f in range(warm):
        hz.frame(scene.common_settings(api, data[f], f, reset=(f == 0)), planes[f], st)
    t0 = time.perf_counter()
    for f in range(warm, warm + frames):
        hz.frame(scene.common_settings(api, data[f], f), planes[f], st)
    dt = time.perf_counter() - t0
    hz.nrd.destroy()
    return w * h * frames / dt / 1e6


def cpu_baseline(pkg, denoiser_names, settings_of, device, w, h):
    """Oracle (scalar C++ port of the same passes, oracle/) on a bounded sample of the same workload (SURVEY.md 8d / BASELINE.md 4): all
    hardware threads at the workload's own frame size (1 warm-up + 2 timed frames: `value`), and - VERDICT r5 item 7 - ONE thread and all
    threads on the SAME smaller frame (1920 x 1080 of the same pipeline, 1 warm-up + 2 timed frames each; a single thread needs ~30 s per
    4K frame) with their ratio. The oracle hands rows out in small chunks to persistent workers (oracle/orc_core.cpp pool_run). Frames are
    rendered on the GPU and copied to the host."""
    if not os.path.exists(graft.ORACLE_LIB):
        return None
    orc = graft.oracle_backend()  # the only use of oracle/ outside tests/ and smoke(): the reported CPU baseline
    cores = min(os.cpu_count() or 1, h)
    warm, frames = 1, 2
    multi = cpu_run(pkg, orc, denoiser_names, settings_of, device, w, h, warm, frames, cores)
    sw, sh = min(w, 1920), min(h, 1080)
    scores = min(os.cpu_count() or 1, sh)
    single = cpu_run(pkg, orc, denoiser_names, settings_of, device, sw, sh, warm, frames, 1)
    same = cpu_run(pkg, orc, denoiser_names, settings_of, device, sw, sh, warm, frames, scores)
    return {"value": round(multi, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "%dx%d (the workload's frame), %d frames after %d warm-up, same pipeline (%s), oracle/ rows in chunks over %d persistent threads"
                      % (w, h, frames, warm, "+".join(denoiser_names), cores),
            "same_size": {"frame": "%dx%d" % (sw, sh), "frames": frames, "warmup": warm, "unit": "Mpixels/s",
                          "one_thread": round(single, 4), "all_threads": round(same, 3), "threads": scores,
                          "ratio": round(same / single, 1) if single > 0 else None},
            "single_thread": {"value": round(single, 4), "unit": "Mpixels/s", "cores": 1,
                              "sample": "%dx%d instance of the same pipeline, %d frames after %d warm-up, one thread" % (sw, sh, frames, warm)}}


OPTIONS = {"checkerboard": False, "atrous": 0}

# Algorithmic bytes per pixel per frame by the CONTRACT's rule (SURVEY.md 8d / BASELINE.md 3: every plane a pass binds counted once
# at the storage size the reference's formats imply - 8-byte guides). The as-built figure the dispatch table reports is larger
# (this build's 16-byte pre-decoded guide texel: 408 instead of 352 for REBLUR_DIFFUSE_SPECULAR); both are reported.
CONTRACT_BPP = {"REFERENCE": 48.0, "SIGMA_SHADOW_TRANSLUCENCY": 76.0, "REBLUR_DIFFUSE": 224.0, "REBLUR_DIFFUSE_SPECULAR": 352.0,
                "RELAX_DIFFUSE_SPECULAR_SH": 770.0}


def contract_bpp(den_names):
    if any(n not in CONTRACT_BPP for n in den_names):
        return None
    total = sum(CONTRACT_BPP[n] for n in den_names)
    if OPTIONS["atrous"] and "RELAX_DIFFUSE_SPECULAR_SH" in den_names:
        total += 76.0 * (OPTIONS["atrous"] - 5)
    return total


def settings_of(api, scene, dens):
    s = {}
    for d in dens:
        if d.name.startswith("REBLUR"):
            # the sample's operating point (Source/NRDSample.cpp:563-585): material-aware filtering on, clamp sigma 1.5
            s[d] = api.ReblurSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1, fastHistoryClampingSigmaScale=1.5,
                                      maxAccumulatedFrameNum=30, maxFastAccumulatedFrameNum=6, maxStabilizedFrameNum=30)
        elif d.name.startswith("RELAX"):
            s[d] = api.RelaxSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1, fastHistoryClampingSigmaScale=1.5)
        elif d.name.startswith("SIGMA"):
            s[d] = api.SigmaSettings(lightDirection=list(scene.sun))
        else:
            s[d] = api.ReferenceSettings()
        if OPTIONS["checkerboard"] and (d.name.startswith("REBLUR") or d.name.startswith("RELAX")):
            s[d].checkerboardMode = int(api.CheckerboardMode.WHITE)
        if OPTIONS["atrous"] and d.name.startswith("RELAX"):
            s[d].atrousIterationNum = OPTIONS["atrous"]
    return s


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    pkg = graft.load_package()
    api, synth, harness = pkg.api, pkg.synth, pkg.harness
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the denoiser passes are HIP kernels, there is no CPU fallback")
    # one rank per GPU over RCCL. NRD_BENCH_DRYRUN_BACKEND=gloo is a plumbing check only (ranks share the GPUs that exist,
    # rows travel through the host): it exercises the N > 1 code path on a 1-GPU box, its numbers mean nothing.
    backend = os.environ.get("NRD_BENCH_DRYRUN_BACKEND", "nccl")
    local = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = os.environ.get("NRD_BENCH_DEVICE") or "cuda:%d" % local  # (NRD_BENCH_DEVICE=cpu: tests/test_bench_flow.py drives this flow on the emulated kernels)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        # a stuck exchange should end the run with an error within minutes, not sit on the node for the default half hour
        if args.native_note is not None and os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
            # this process replaced one that abandoned the native tiler (--tiler auto below). Under torchrun the key-value store lives in the
            # launcher and still holds the abandoned run's rendezvous keys: the same store, behind a prefix of its own
            store = dist.PrefixStore("python_tiler_restart", dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, is_master=False,
                                                                           timeout=datetime.timedelta(seconds=300)))
            dist.init_process_group(backend, store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        # first communication of the group is one every rank takes part in (batched P2P between row neighbours comes later)
        hello = torch.ones(1, device=dev)
        dist.all_reduce(hello)
        torch.cuda.synchronize()
        if int(hello.item()) != world:
            raise SystemExit("all_reduce over %d ranks returned %s" % (world, hello.item()))

    OPTIONS["checkerboard"], OPTIONS["atrous"] = args.checkerboard, args.atrous
    if args.checkerboard and (world > 1 or args.force_tiled):
        raise SystemExit("--checkerboard is wired into the single-GPU runner only")
    w, wl_h, den_names = WORKLOADS[args.workload]
    dens = [api.Denoiser[n] for n in den_names]
    hip = pkg.hip_backend(dev)
    if args.workload == "sample_passes_4k":
        if world > 1:
            raise SystemExit("sample_passes_4k is a single-GPU workload")
        print(json.dumps(run_sample_passes(args, pkg, hip, dev, w, wl_h)))
        return
    strong = args.scaling == "strong"
    band_h = wl_h  # rows a rank owns in weak mode / at N = 1; strong mode: band_layout() splits wl_h

    if world == 1 and not args.force_tiled:
        from nrd_sample_amd.harness import Harness

        preset, scene_kw = None, {}
        if args.preset:
            from nrd_sample_amd import sample_tests

            path, _, index = args.preset.rpartition(":")
            preset = sample_tests.load_presets(path)[int(index)]
            scene_kw = sample_tests.scene_kwargs(preset)
        scene = synth.Scene(w, band_h, dolly=args.dolly, device=dev, denoiser="RELAX" if den_names[0].startswith("RELAX") else "REBLUR",
                            roll_deg=args.roll, **scene_kw)
        hz = Harness(hip, dens, w, band_h, separate_passes=args.separate_passes)
        st = settings_of(api, scene, dens) if preset is None else sample_tests.denoiser_settings(api, preset, scene, dens, first_frame=False)
        runner = SingleRunner(api, hz, scene, dens, args.unique_frames, st)
        frame_h = band_h
    else:
        from nrd_sample_amd.tiler import TiledRunner

        frame_h = wl_h if strong else wl_h * world

        def make_runner(kind):
            return TiledRunner(pkg, hip, dev, dens, w, frame_h, rank, world, args.unique_frames, args.dolly, settings_of,
                               tiler=kind, motion_rows=args.motion_rows, balance=strong and not args.even_bands)

        if args.tiler == "auto":
            # The value of record should come from the C++ tiler (ncclSend / ncclRecv groups below the C-ABI). Its RCCL calls have executed on
            # one GPU only (tests/test_rccl_loopback.py), so it is tried under two guards: an exception on ANY rank sends every rank to the
            # Python tiler (the ranks agree through one all_reduce), and a probe that does not come back within --native-deadline makes every
            # rank replace itself with the same command + --tiler python (a hung collective cannot be cancelled from inside the process).
            import threading

            def restart_with_python_tiler():
                note = "native tiler: set-up + two-frame probe not finished after %g s on rank %d" % (args.native_deadline, rank)
                sys.stderr.write(note + " - restarting with --tiler python\n")
                sys.stderr.flush()
                # a fresh rendezvous: the ranks replace themselves milliseconds apart, and a late one must not find the key-value store of the
                # abandoned run still answering on the old port (rank 0 hosts it) - under torchrun the launcher hosts it: the port stays and
                # the restarted ranks meet behind a key prefix instead (init_process_group above)
                if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") != "True" and os.environ.get("MASTER_PORT", "").isdigit():
                    os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 1)
                os.execv(sys.executable, [sys.executable] + sys.argv + ["--tiler", "python", "--native-note", note])

            guard = threading.Timer(args.native_deadline, restart_with_python_tiler)
            guard.daemon = True
            guard.start()
            err = None
            try:
                if os.environ.get("NRD_BENCH_NATIVE_FAIL_RANK") in (str(rank), "all"):  # (tests/test_bench_flow.py: the fallback itself is tested)
                    raise RuntimeError("injected failure of the native tiler on rank %d" % rank)
                runner = make_runner("native")
                for f in range(2):
                    runner.step(f, reset=(f == 0))
                runner.finish()
                torch.cuda.synchronize() if dev != "cpu" else None
            except Exception as e:  # (reported in config.native_tiler)
                err = "%s: %s" % (type(e).__name__, e)
            if world > 1:
                okf = torch.tensor([0.0 if err else 1.0], device=dev)
                dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                if float(okf.item()) == 0.0 and err is None:
                    err = "another rank's native tiler raised"
            guard.cancel()
            if err is None:
                args.tiler = "native"
            else:
                args.native_note = err
                args.tiler = "python"
                runner = None
                torch.cuda.empty_cache() if dev != "cpu" else None
                runner = make_runner("python")
        else:
            runner = make_runner(args.tiler)
        band_h = runner.band.layout["own_rows"]

    def accum_length(runner):
        return max([int(getattr(st, "maxAccumulatedFrameNum", 0)) for st in runner.settings.values()] +
                   [int(getattr(st, "diffuseMaxAccumulatedFrameNum", 0)) for st in runner.settings.values()] + [0])

    def preroll_frames(runner):
        """untimed frames in FRONT of the --warmup frames, so that the timed region is the steady state whatever --warmup says: the blur
        radii shrink with the history length, which saturates after maxAccumulatedFrameNum frames (30 at the sample's operating point) -
        a run timed after 5 warm-up frames measures wider, slower blurs than the denoiser does from frame 32 on (round 3: 7862 vs 8457)"""
        return 0 if args.no_preroll else max(0, accum_length(runner) + 2 - args.warmup)

    def timed_run(runner):
        """pre-roll to the steady state, W untimed warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; returns
        the wall time of the timed region (max over ranks)"""
        import gc

        pre = preroll_frames(runner)
        # no cyclic garbage collection inside a timed region: a generation-2 pass of a torch-sized heap stops the launch thread for ~35 ms
        # (measured round 5, tools/hwt_leg_probe.py: one HistoryFix event pair 22 ms wide, the next frames 10 % slower while the idled
        # GPU's clocks ramp up again) - it struck whichever leg happened to cross the allocation threshold
        gc.collect()
        gc.disable()
        try:
            return _timed_run(runner, pre)
        finally:
            gc.enable()

    def _timed_run(runner, pre):
        for f in range(pre + args.warmup):
            runner.step(f, reset=(f == 0))
        if hasattr(runner, "finish"):
            runner.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        # timed region: exactly K steps, HIP events around every dispatch of every S-th step on the launch stream
        stride = max(args.event_stride, 1)
        t0 = time.perf_counter()
        for f in range(pre + args.warmup, pre + args.warmup + args.steps):
            runner.enable_events((f - pre - args.warmup) % stride == 0)
            runner.step(f, reset=False)
        if hasattr(runner, "finish"):
            runner.finish()  # row tiler: halo rows of the last frame's permanent planes still travelling
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        runner.local_ms_per_step = dt / args.steps * 1e3  # this rank's own wall time (the value of record is the max over ranks)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    ids = [int(d) for d in dens]
    dt = timed_run(runner)
    per_pass = runner.pass_times_ms()  # {name: (avg ms, bytes per pixel)}

    # ---- N > 1 extras (outside the timed region of record): per-rank GPU time, the native C++ / RCCL tiler, bit identity ----
    rank_ms = native_leg = identical = None
    tiled = world > 1 or args.force_tiled
    if world > 1:
        # GPU-busy ms per frame of this rank: the sum of its dispatch times; the C++ tiler steps the dispatches itself (no per-dispatch events):
        # its ranks report their wall time per frame instead
        mine = sum(v[0] for v in per_pass.values()) if per_pass else getattr(runner, "local_ms_per_step", float("nan"))
        mt = torch.tensor([mine], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(mt) for _ in range(world)]
        dist.all_gather(gathered, mt)
        rank_ms = [round(float(g.item()), 4) for g in gathered]  # GPU-busy ms per frame of every rank (sum of its dispatch times)
    if rank == 0:
        pixels_band = w * band_h
        total_pixels = w * frame_h * args.steps
        value = total_pixels / dt / 1e6
        ms_per_step = dt / args.steps * 1e3
        max_accum, pre = accum_length(runner), preroll_frames(runner)
        state = ("steady state (accumulation saturated%s)" % (": %d untimed pre-roll frames in front of the %d warm-up frames" % (pre, args.warmup) if pre else "")
                 if pre + args.warmup >= max_accum else "warm-up %d frames (accumulation saturates at %d)" % (args.warmup, max_accum))
        tiled_desc = "" if world == 1 and not args.force_tiled else " row-tiled %d x ~%d rows (%s scaling), halo %d rows, %s tiler" % (
            world, band_h, args.scaling, runner.halo, args.tiler)
        bpp_contract = contract_bpp(den_names)
        out = {
            "metric": "Mpixels/s full %s diff+spec pipeline" % den_names[0].split("_")[0], "value": round(value, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak" if (world == 1 or not strong) else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s, %dx%d%s, %s" % (args.workload, "+".join(den_names), w, frame_h, tiled_desc, state) +
                       (" [camera rolled %g deg: layout probe]" % args.roll if args.roll else "") +
                       (" [recorded preset %s]" % os.path.basename(args.preset) if args.preset else ""),
                       "unique_input_frames": args.unique_frames, "storage_dtype": "f16 planes (f32 viewZ), f32 arithmetic"},
        }
        if per_pass:
            dom = max(per_pass.items(), key=lambda kv: kv[1][0])
            dom_name, (dom_ms, dom_bpp) = dom
            achieved = dom_bpp * pixels_band / (dom_ms * 1e-3) / 1e9
            sum_ms = sum(v[0] for v in per_pass.values())
            sum_bpp = sum(v[1] for v in per_pass.values())
            out["config"]["algorithmic_bytes_per_pixel"] = round(sum_bpp, 2)
            out["roofline"] = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args.workload, dom_name),
                               # a live run cannot collect counters: the figure comes from a committed rocprofv3 --pmc table - which, and of which build
                               "traffic_source": traffic_source(args.workload, pkg.HIP_LIB),
                               # whole pass chain on both denominators: the as-built plane layout and the contract's rule
                               "algorithmic_bytes_as_built": round(sum_bpp, 2), "pipeline_frac": round(sum_bpp * pixels_band / (sum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "algorithmic_bytes_contract": bpp_contract,
                               "pipeline_frac_contract": None if bpp_contract is None else round(bpp_contract * pixels_band / (sum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            out["passes_ms"] = {k: round(v[0], 4) for k, v in per_pass.items()}
            if hasattr(runner, "sky_fraction"):
                # the sky-proof figures: no kernel moves the bytes of a pixel without geometry, so the fractions above
                # flatter a frame that shows sky. *_geometry counts the algorithmic bytes of geometry pixels only ((1 - sky_fraction) x bytes).
                # The figure to hold against the 70 % target is config.full_coverage (the same scene with no sky in view): BASELINE.md 3
                geo = 1.0 - runner.sky_fraction()
                r = out["roofline"]
                r["frac_geometry"] = round(r["frac"] * geo, 4)
                r["pipeline_frac_geometry"] = round(r["pipeline_frac"] * geo, 4)
                r["pipeline_frac_contract_geometry"] = None if r["pipeline_frac_contract"] is None else round(r["pipeline_frac_contract"] * geo, 4)
        else:  # the C++ tiler steps the dispatches itself: whole-frame figures only
            sum_bpp = sum(b for _, b in runner.dispatch_table())
            out["config"]["algorithmic_bytes_per_pixel"] = round(sum_bpp, 2)
            out["roofline"] = {"bound": "hbm", "kernel": "whole frame (no per-dispatch events with --tiler native)", "achieved": round(sum_bpp * pixels_band / (ms_per_step * 1e-3) / 1e9, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sum_bpp * pixels_band / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                               "algorithmic_bytes_as_built": round(sum_bpp, 2), "algorithmic_bytes_contract": bpp_contract}
        if hasattr(runner, "sky_fraction"):
            # what the scene costs depends on how much of it is geometry: sky pixels (|viewZ| > denoisingRange) leave every pass after
            # one guide load. The headline scene shows sky above its back wall; the same pipeline is timed once more on the same scene
            # with the wall closing the view (no sky pixel) and reported beside it.
            out["config"]["sky_fraction"] = round(runner.sky_fraction(), 4)
        if hasattr(runner, "tiler"):
            t = runner.tiler
            out["config"]["halo_exchange_bytes_per_frame_rank0"] = int(t.bytes_exchanged / max(preroll_frames(runner) + args.warmup + args.steps, 1))
            # what the exchange would cost if NOTHING of it overlapped with compute: rank 0 has one neighbour; an interior rank of N >= 3
            # moves the same volume over EACH of its two links, both directions at once. xGMI: ~50-75 GB/s per direction per link in
            # practice (MI355X guide: 153 GB/s bidirectional peak per link); DESIGN.md 7 has the per-pass table behind this figure
            per_neighbour = out["config"]["halo_exchange_bytes_per_frame_rank0"] + int(getattr(t, "input_bytes_per_frame", 0))
            out["config"]["predicted_exchange_ms"] = {"bytes_per_neighbour_per_frame": per_neighbour,
                                                      "at_50_GBs_per_direction": round(per_neighbour / 50e9 * 1e3, 3),
                                                      "at_75_GBs_per_direction": round(per_neighbour / 75e9 * 1e3, 3),
                                                      "band_compute_ms_rank0": round(sum(v[0] for v in per_pass.values()), 4) if per_pass else None}
            if per_pass and hasattr(t, "_plan"):
                # the same bytes dispatch by dispatch against the measured interior each exchange can hide behind (strips-first schedule):
                # what the frame SHOULD cost on real links - the figure a SCALE run is to be compared with (nrd-sample_amd/tiler.py)
                try:
                    from nrd_sample_amd.tiler import exchange_overlap_model, plan_bytes

                    disp = runner.band.nrd.dispatches(ids)
                    planes = {(pool << 16) | i: p["bpt"] for pool in (0, 1) for i, p in enumerate(runner.band.nrd.pools[pool])}
                    pb = plan_bytes(t._plan(ids, disp), planes, w)
                    pm = [per_pass[d["name"]][0] for d in disp]
                    nb = 1 if world <= 2 else 2
                    out["config"]["exchange_overlap_model"] = {
                        "%d_GBs_%d_us" % (gbs, lat): {k: v for k, v in exchange_overlap_model([d["name"] for d in disp], pm, pb, band_h, nb, gbs, lat).items() if k != "per_dispatch" or (gbs, lat) == (50, 20)}
                        for gbs, lat in ((50, 20), (75, 10), (35, 50))}
                    out["config"]["exchange_overlap_model"]["note"] = ("rank 0's measured dispatch times (ms, strips + interior) and plan bytes; an interior band (2 neighbours) "
                                                                       "for N > 2; xGMI rate per direction per link and end-to-end latency per exchange group are ASSUMED")
                except Exception as e:
                    out["config"]["exchange_overlap_model"] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["config"]["band_rows"] = [b1 - b0 for b0, b1 in zip(runner.band.bounds, runner.band.bounds[1:])]
            out["config"]["band_split"] = "cost-balanced (geometry tiles + 0.15 x sky tiles of the first frame)" if runner.bounds else "even tile rows"
        if rank_ms is not None:
            out["config"]["rank_ms"] = rank_ms

    # ---- N > 1 extras, every rank takes part: the C++ / RCCL tiler leg and the bit-identity check. Both run code whose transport has
    # never executed on real multi-GPU hardware; an exception is caught below, and a HANG (a receive that never completes) is bounded by
    # a watchdog: after --extras-deadline seconds rank 0 prints the line of record as it stands - the value above is complete - with the
    # unfinished extras marked, and every rank leaves the process. Whatever goes wrong there must not cost the line.
    import threading

    printed = threading.Lock()
    watchdog = None
    if tiled and (not args.no_native_leg or not args.no_identity_check):
        def give_up():
            if rank == 0 and printed.acquire(blocking=False):
                cfg = out["config"]
                cfg.setdefault("native_tiler", native_leg if native_leg is not None else {"error": "not finished after %d s: skipped by the watchdog" % args.extras_deadline})
                cfg.setdefault("tiled_bit_identical", None)
                cfg.setdefault("tiled_bit_identical_detail", "not finished after %d s: skipped by the watchdog" % args.extras_deadline)
                print(json.dumps(out), flush=True)
            os._exit(0)

        watchdog = threading.Timer(args.extras_deadline, give_up)
        watchdog.daemon = True
        watchdog.start()
    if tiled and args.tiler == "python" and not args.no_native_leg and args.native_note is None:  # (--tiler python given explicitly: the C++ tiler as a second leg)
        # The value of record is the Python tiler's (its transport, PyTorch's RCCL binding, is the proven one). The C++ tiler below
        # the C-ABI (ncclSend / ncclRecv groups on a side stream) runs the same workload afterwards, reported beside it; whatever
        # goes wrong there must not cost the line above.
        try:
            from nrd_sample_amd.tiler import TiledRunner

            native = TiledRunner(pkg, hip, dev, dens, w, frame_h, rank, world, args.unique_frames, args.dolly, settings_of, tiler="native",
                                 motion_rows=args.motion_rows, balance=strong and not args.even_bands)
            dt_n = timed_run(native)
            native_leg = {"value": round(w * frame_h * args.steps / dt_n / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(dt_n / args.steps * 1e3, 4),
                          "transport": "rccl" if backend == "nccl" else "caller callbacks over torch.distributed (%s)" % backend,
                          "halo_exchange_bytes_per_frame_rank0": int(native.tiler.bytes_exchanged / max(preroll_frames(native) + args.warmup + args.steps, 1))}
            del native
            torch.cuda.empty_cache()
        except Exception as e:
            native_leg = {"error": "%s: %s" % (type(e).__name__, e)}
    if tiled and strong and not args.no_identity_check:
        try:
            from nrd_sample_amd.tiler import verify_tiled_against_single

            ok, detail = verify_tiled_against_single(pkg, hip, dev, dens, w, frame_h, rank, world, settings_of, args.dolly, runner.halo,
                                                     runner.bounds, tiler=args.tiler)
            identical = {"identical": ok, "detail": detail}
        except Exception as e:
            identical = {"identical": None, "detail": "%s: %s" % (type(e).__name__, e)}
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        if not printed.acquire(blocking=False):  # the watchdog got there first
            os._exit(0)
        if native_leg is not None:
            out["config"]["native_tiler"] = native_leg
        elif tiled:
            out["config"]["native_tiler"] = {"used_for_value": args.tiler == "native", "transport": "rccl" if backend == "nccl" else "caller callbacks over torch.distributed (%s)" % backend,
                                             "fallback_reason": args.native_note}
        if identical is not None:
            out["config"]["tiled_bit_identical"] = identical["identical"]
            out["config"]["tiled_bit_identical_detail"] = identical["detail"]
        if world == 1 and not args.force_tiled and not args.no_full_coverage and not args.preset:
            try:
                del runner, hz
                torch.cuda.empty_cache()
                scene_fc = synth.Scene(w, band_h, dolly=args.dolly, device=dev, denoiser="RELAX" if den_names[0].startswith("RELAX") else "REBLUR",
                                       roll_deg=args.roll, wall_height=float("inf"))
                hz_fc = Harness(hip, dens, w, band_h, separate_passes=args.separate_passes)
                runner_fc = SingleRunner(api, hz_fc, scene_fc, dens, args.unique_frames, settings_of(api, scene_fc, dens))
                dt_fc = timed_run(runner_fc)
                pp = runner_fc.pass_times_ms()
                sum_ms_fc = sum(v[0] for v in pp.values())
                fc = {"value": round(w * frame_h * args.steps / dt_fc / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(dt_fc / args.steps * 1e3, 4),
                      "sky_fraction": round(runner_fc.sky_fraction(), 4),
                      "pipeline_frac_contract": None if bpp_contract is None else round(bpp_contract * w * frame_h / (sum_ms_fc * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      # the slowest kernel of this leg on its own algorithmic bytes (the sky-free counterpart of roofline.frac)
                      "roofline_frac": round(max(v[1] * w * frame_h / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS for v in [max(pp.values(), key=lambda t: t[0])]), 4),
                      "roofline_kernel": max(pp.items(), key=lambda kv: kv[1][0])[0],
                      "passes_ms": {k: round(v[0], 4) for k, v in pp.items()},
                      "scene": "the same scene with the back wall closing the view (no sky pixel), same settings, steps and warm-up"}
                out["config"]["full_coverage"] = fc
                del runner_fc, hz_fc
                torch.cuda.empty_cache()
            except Exception as e:  # a reported extra: never lose the headline line to it
                out["config"]["full_coverage"] = {"error": str(e)}
        if world == 1 and not args.force_tiled and not args.no_frozen_leg and not args.preset and os.path.exists(pkg.HIP_LIB_FROZEN):
            # What the cheaper formulas of rounds 1-3 would buy, and how far their output is from the default's: the same workload through
            # libnrdhip_frozen.so (csrc/nrd_device.h NRD_UPSTREAM_FORMULAS = 0: compact-support hit-distance weight instead of exp(-3|x|),
            # normal weight on the squared angle instead of the angle, Blur rotation per 2x2 quad instead of per pixel, RELAX in YCoCg inside)
            try:
                hip_fz = pkg.hip_backend(dev, flavour="frozen")
                scene_fz = synth.Scene(w, band_h, dolly=args.dolly, device=dev, denoiser="RELAX" if den_names[0].startswith("RELAX") else "REBLUR",
                                       roll_deg=args.roll)
                hz_fz = Harness(hip_fz, dens, w, band_h, separate_passes=args.separate_passes)
                runner_fz = SingleRunner(api, hz_fz, scene_fz, dens, args.unique_frames, settings_of(api, scene_fz, dens))
                dt_fz = timed_run(runner_fz)
                pp = runner_fz.pass_times_ms()
                leg = {"value": round(w * frame_h * args.steps / dt_fz / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(dt_fz / args.steps * 1e3, 4),
                       "passes_ms": {k: round(v[0], 4) for k, v in pp.items()},
                       "what": "same workload, libnrdhip_frozen.so: hit-distance weight (1-|x|)^2 instead of exp(-3|x|), normal weight on the squared angle "
                               "instead of the angle (upstream's AcosApprox: the chord of the two normals), Blur rotation per 2x2 quad instead of per pixel, RELAX in YCoCg inside instead of linear RGB"}
                del runner_fz, hz_fz
                torch.cuda.empty_cache()
                out["config"]["frozen_formulas"] = leg
                leg["distance_from_default"] = flavour_distance(pkg, api, synth, Harness, hip, hip_fz, dev, dens, den_names, args)
            except Exception as e:
                out["config"].setdefault("frozen_formulas", {})["error"] = str(e)
        if world == 1 and not args.force_tiled and not args.no_frozen_leg and not args.preset and os.path.exists(pkg.HIP_LIB_HWT):
            # The OPTIONAL flavour with the GPU's transcendental instructions in the weight arithmetic of the spatial filters (libnrdhip_hwt.so,
            # csrc/nrd_device.h NRD_HW_TRANSCENDENTALS): timed beside the default; not bit-reproducible on a CPU and measured NOT to hold the
            # 1-ULP bar against its checker at isolated pixels (profiles/r05_ab_hw_transcendentals.txt), hence not the library of record
            try:
                hip_hw = pkg.hip_backend(dev, flavour="hwt")
                scene_hw = synth.Scene(w, band_h, dolly=args.dolly, device=dev, denoiser="RELAX" if den_names[0].startswith("RELAX") else "REBLUR", roll_deg=args.roll)
                hz_hw = Harness(hip_hw, dens, w, band_h, separate_passes=args.separate_passes)
                runner_hw = SingleRunner(api, hz_hw, scene_hw, dens, args.unique_frames, settings_of(api, scene_hw, dens))
                dt_hw = timed_run(runner_hw)
                pp = runner_hw.pass_times_ms()
                leg = {"value": round(w * frame_h * args.steps / dt_hw / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(dt_hw / args.steps * 1e3, 4),
                       "passes_ms": {k: round(v[0], 4) for k, v in pp.items()},
                       "what": "same workload, libnrdhip_hwt.so: v_rcp_f32 / v_sqrt_f32 / v_exp_f32 in the weight-class arithmetic of the spatial filters (tap weights, their "
                               "per-pixel parameters, 1 / weight sum); everything a discrete decision hangs on keeps the exact sequences. OPTIONAL: fails the 1-ULP bar at isolated pixels"}
                del runner_hw, hz_hw
                torch.cuda.empty_cache()
                out["config"]["hw_transcendentals"] = leg
                leg["distance_from_default"] = flavour_distance(pkg, api, synth, Harness, hip, hip_hw, dev, dens, den_names, args)
            except Exception as e:
                out["config"].setdefault("hw_transcendentals", {})["error"] = str(e)
        if world == 1 and not args.force_tiled and not args.no_young_leg and not args.preset and not args.no_preroll:
            # the pre-roll is a METHODOLOGY change of round 4 (rounds 1-3 timed right behind the warm-up frames: young histories,
            # wider blur radii, slower frames). The same workload once more without it, so that both regimes stand in one line of record.
            try:
                scene_y = synth.Scene(w, band_h, dolly=args.dolly, device=dev, denoiser="RELAX" if den_names[0].startswith("RELAX") else "REBLUR", roll_deg=args.roll)
                hz_y = Harness(hip, dens, w, band_h, separate_passes=args.separate_passes)
                runner_y = SingleRunner(api, hz_y, scene_y, dens, args.unique_frames, settings_of(api, scene_y, dens))
                args.no_preroll = True
                try:
                    dt_y = timed_run(runner_y)
                finally:
                    args.no_preroll = False
                out["config"]["without_preroll"] = {"value": round(w * frame_h * args.steps / dt_y / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(dt_y / args.steps * 1e3, 4),
                                                    "what": "timed right behind the %d warm-up frames (histories %d..%d frames old of %d: wider blurs) - the regime rounds 1-3 reported"
                                                            % (args.warmup, args.warmup, args.warmup + args.steps, accum_length(runner_y))}
                del runner_y, hz_y
                torch.cuda.empty_cache()
            except Exception as e:
                out["config"]["without_preroll"] = {"error": str(e)}
        if world == 1 and not args.force_tiled and not args.no_graph_leg and not args.preset:
            # NRDHIP_FLAG_GRAPH: the frame as ONE HIP graph launch (captured every frame, the executable graph patched with the new kernel
            # arguments) against pass-by-pass launches, both on the same non-default stream and without per-pass events: on the
            # workload itself and on the band one rank of 8 holds of the 8K frame (BASELINE config 5), where a frame is ~0.5 ms of
            # GPU work and the launch thread matters most
            try:
                leg = {}
                side = torch.cuda.Stream(device=dev)
                for tag, (gw, gh) in (("workload", (w, band_h)), ("band_%dx%d" % GRAPH_LEG_BAND, GRAPH_LEG_BAND)):
                    row = {}
                    for mode in ("direct", "graph"):
                        scene_g = synth.Scene(gw, gh, dolly=args.dolly, device=dev, denoiser="RELAX" if den_names[0].startswith("RELAX") else "REBLUR",
                                              roll_deg=args.roll)
                        hz_g = Harness(hip, dens, gw, gh, graph=(mode == "graph"))
                        runner_g = SingleRunner(api, hz_g, scene_g, dens, args.unique_frames, settings_of(api, scene_g, dens))
                        runner_g.never_events = True
                        torch.cuda.synchronize()
                        with torch.cuda.stream(side):
                            dt_g = timed_run(runner_g)
                        row[mode + "_ms_per_step"] = round(dt_g / args.steps * 1e3, 4)
                        if mode == "graph":
                            row["graph_stats"] = hz_g.nrd.graph_stats()
                        del runner_g, hz_g, scene_g
                        torch.cuda.empty_cache()
                    leg[tag] = row
                out["config"]["graph_replay"] = leg
            except Exception as e:
                out["config"]["graph_replay"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:  # reported on rank 0 at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(pkg, den_names, settings_of, dev, w, wl_h)
            except Exception as e:  # the baseline is a reported extra; never fail the GPU number because of it
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def run_sample_passes(args, pkg, hip, dev, w, h):
    """--workload sample_passes_4k: the sample-side passes either side of the denoiser, each timed on its own with HIP
    events and priced against its algorithmic bytes: ConfidenceBlur x 5 at 1/5 resolution (Source/NRDSample.cpp:3999-4026), the
    front-end pack (TraceOpaque.cs.hlsl:421-801), the back-end unpack + composition (Composition.cs.hlsl:57-188) and TAA
    (Taa.cs.hlsl:11-159). Not the headline metric: one JSON line of the same shape, `value` = pixels / time of the whole chain."""
    import numpy as np
    import torch

    from nrd_sample_amd import sample_passes as sp

    g = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda *shape: torch.rand(*shape, device=dev, generator=g, dtype=torch.float32)
    as_bytes = lambda t: t.contiguous().view(torch.uint8).reshape(t.shape[0], -1)
    n = torch.nn.functional.normalize(rnd(h, w, 3) - 0.5, dim=-1)
    normal = as_bytes(torch.cat([n, rnd(h, w, 1)], -1))
    mat = as_bytes(torch.floor(rnd(h, w) * 3.99))
    viewz = as_bytes(1.0 + 30.0 * rnd(h, w))
    rad = lambda: as_bytes(torch.cat([rnd(h, w, 3), 5.0 * rnd(h, w, 1)], -1))
    diff32, spec32, shadow32 = rad(), rad(), as_bytes(torch.cat([70000.0 * (rnd(h, w, 1) > 0.5) + rnd(h, w, 1), rnd(h, w, 3)], -1))
    z8 = lambda bpt: torch.zeros((h, w * bpt), dtype=torch.uint8, device=dev)
    nr, pdiff, pspec, pen, transl = z8(4), z8(8), z8(8), z8(2), z8(4)
    udiff, uspec, ushadow, cdiff, cspec = z8(8), z8(8), z8(8), z8(8), z8(8)
    base = as_bytes((rnd(h, w, 4) * 255).to(torch.uint8))
    mv = as_bytes(torch.cat([rnd(h, w, 2) - 0.5, torch.zeros(h, w, 1, device=dev), 0.125 * (1.0 + 30.0 * rnd(h, w, 1))], -1).to(torch.float16))
    hist_a, hist_b = as_bytes(rnd(h, w, 4).to(torch.float16)), z8(8)
    sw, sh_ = sp.sharc_dims(w, h)
    ping = torch.from_numpy(sp.synth_gradient(sw, sh_).view(np.uint8).reshape(sh_, sw * 8)).to(dev)
    pong = torch.zeros_like(ping)
    frustum = (-1.0, 0.5625, 2.0, -1.125)
    hdp = (3.0, 0.1, 20.0, -25.0)
    stages = [
        ("Sample::ConfidenceBlur x5 (1/25 of the pixels)", 5 * 16.0 * sw * sh_ / (w * h),
         lambda: sp.confidence_blur(hip, ping, pong, sw, sh_, frustum, float(w), 1.0 / (0.5 * h), 0, 30)),
        ("Sample::FrontEndPack", 72.0 + 26.0,
         lambda: sp.frontend_pack(hip, w, h, hit_distance_parameters=hdp, tan_of_light_angular_radius=0.005, normal=normal, material_id=mat, viewz=viewz,
                                  diff=diff32, spec=spec32, shadow=shadow32, out_normal_roughness=nr, out_diff=pdiff, out_spec=pspec,
                                  out_penumbra=pen, out_translucency=transl)),
        ("Sample::BackEndUnpack", 20.0 + 24.0,
         lambda: sp.backend_unpack(hip, w, h, diff=pdiff, spec=pspec, shadow=transl, out_diff=udiff, out_spec=uspec, out_shadow=ushadow)),
        ("Sample::Compose", 28.0 + 16.0,
         lambda: sp.compose(hip, w, h, diff=udiff, spec=uspec, normal_roughness=nr, viewz=viewz, base_color_metalness=base, out_diff=cdiff, out_spec=cspec)),
        ("Sample::Taa", 24.0 + 8.0, lambda: sp.taa(hip, mv, cdiff, hist_a, hist_b, w, h)),
    ]
    for _, _, fn in stages:
        for _ in range(max(args.warmup // 8, 2)):
            fn()
    torch.cuda.synchronize()
    times = {}
    t0 = time.perf_counter()
    for name, bpp, fn in stages:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        times[name] = (a.elapsed_time(b) / args.steps, bpp)
    dt = time.perf_counter() - t0
    total_ms = sum(v[0] for v in times.values())
    dom = max(times.items(), key=lambda kv: kv[1][0])
    gbs = {k: round(bpp * w * h / (ms * 1e-3) / 1e9, 1) for k, (ms, bpp) in times.items()}
    return {"metric": "Mpixels/s sample-side passes (pack, unpack, compose, TAA, ConfidenceBlur)", "value": round(w * h / (total_ms * 1e-3) / 1e6, 2),
            "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(total_ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "sample_passes_4k: the sample's passes around the denoiser at %dx%d (ConfidenceBlur at %dx%d); NOT the headline metric"
                                   % (w, h, sw, sh_), "wall_s": round(dt, 3)},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": gbs[dom[0]], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs[dom[0]] / HBM_PEAK_GBS, 4), "traffic": None},
            "passes_ms": {k: round(v[0], 4) for k, v in times.items()}, "passes_gbs": gbs,
            "passes_algorithmic_bytes_per_pixel": {k: round(v[1], 2) for k, v in times.items()}}


def pingpong(n, f):
    """camera path position of step f: 0,1,..,n-1,n-2,..,1,0,1,.. (consistent motion vectors in both directions)"""
    if n == 1:
        return 0
    period = 2 * (n - 1)
    k = f % period
    return k if k < n else period - k


def flavour_distance(pkg, api, synth, Harness, hip_a, hip_b, dev, dens, den_names, args):
    """PSNR and the share of values more than 1 fp16 ULP apart between the outputs of two build flavours of the library, per OUT_* plane,
    after `frames` frames (>= the accumulation length) of the same denoisers at 1920x1080 on the bench scene"""
    import numpy as np
    import torch

    w, h, frames = FLAVOUR_DISTANCE
    outs = []
    for hip_x in (hip_a, hip_b):
        scene = synth.Scene(w, h, dolly=args.dolly, device=dev, denoiser="RELAX" if den_names[0].startswith("RELAX") else "REBLUR", roll_deg=args.roll)
        hz = Harness(hip_x, dens, w, h)
        runner = SingleRunner(api, hz, scene, dens, args.unique_frames, settings_of(api, scene, dens))
        for f in range(frames):
            runner.step(f, reset=(f == 0))
        torch.cuda.synchronize()
        outs.append({k: np.asarray(hz.fetch(v)).copy() for k, v in hz.outputs.items()})
        del runner, hz
        torch.cuda.empty_cache()
    res = {"frames": frames, "size": "%dx%d" % (w, h)}
    for key in ("out_diff", "out_spec"):
        a16, b16 = outs[0][key].view(np.float16), outs[1][key].view(np.float16)
        if not a16.any() and not b16.any():
            continue
        a, b = a16.astype(np.float64), b16.astype(np.float64)
        mse = float(np.mean((a - b) ** 2))
        peak = max(float(np.abs(a).max()), 1e-12)
        ia, ib = a16.view(np.int16).astype(np.int32), b16.view(np.int16).astype(np.int32)
        oa, ob = np.where(ia < 0, -32768 - ia, ia), np.where(ib < 0, -32768 - ib, ib)
        res[key] = {"psnr_db": None if mse == 0 else round(10.0 * np.log10(peak * peak / mse), 2), "ulp_gt1_frac": round(float((np.abs(oa - ob) > 1).mean()), 4)}
    return res


class SingleRunner:
    """one GPU: the sample's per-frame call sequence through nrd-sample_amd/harness.py, dispatch by dispatch.
    `unique` frames along a camera dolly are generated on the device once and replayed forwards then backwards; every frame
    carries two motion-vector planes (previous camera = the neighbour on either side) so reprojection is always consistent."""

    def __init__(self, api, hz, scene, dens, unique, settings):
        import torch

        self.api, self.hz, self.scene, self.dens, self.settings = api, hz, scene, dens, settings
        self.ids = [int(d) for d in dens]
        self.unique = unique
        self.frames = []
        for i in range(unique):
            fwd = scene.frame(i, prev_index=max(i - 1, 0))
            bwd = scene.frame(i, prev_index=min(i + 1, unique - 1))
            planes = hz.upload(fwd)
            mv_b = hz.upload({"mv": bwd["mv"]})["mv"]
            rec = dict(planes=planes, mv_f=planes["mv"], mv_b=mv_b, fwd=fwd, bwd=bwd)
            if OPTIONS["checkerboard"]:  # the squares carrying a signal alternate with the parity of frameIndex
                from nrd_sample_amd.harness import to_checkerboard
                rec["cb"] = [hz.upload(to_checkerboard(fwd, par, white=True)) for par in (0, 1)]
            self.frames.append(rec)
        torch.cuda.synchronize()
        self.events_on = False
        self.events = []
        self.names = None

    def enable_events(self, on):
        self.events_on = on and not getattr(self, "never_events", False)

    def sky_fraction(self):
        """share of the frame's pixels beyond the denoising range (first input frame; the dolly moves it by a fraction of a percent)"""
        import torch

        fr = self.frames[0]["fwd"]
        z = fr["viewz"]
        z = z if hasattr(z, "abs") and hasattr(z, "float") else torch.as_tensor(z)
        rng = float(self.scene.common_settings(self.api, fr, 0).denoisingRange)
        return float((z.abs() > rng).float().mean().item())

    def step(self, f, reset):
        import torch

        hz, api = self.hz, self.api
        cur = pingpong(self.unique, f)
        prev = pingpong(self.unique, f - 1) if f > 0 else min(1, self.unique - 1)
        fr = self.frames[cur]
        backward = prev > cur
        planes = dict(fr["cb"][f & 1] if "cb" in fr else fr["planes"])
        planes["mv"] = fr["mv_b"] if backward else fr["mv_f"]
        cs = self.scene.common_settings(api, fr["bwd"] if backward else fr["fwd"], f, reset=reset)
        hz.nrd.new_frame()
        hz.nrd.set_common_settings(cs)
        hz.bind(planes)
        for d in self.dens:
            hz.nrd.set_denoiser_settings(int(d), self.settings[d])
        if not self.events_on:
            hz.nrd.denoise(self.ids)
            return
        if self.names is None:
            info = hz.nrd.dispatches(self.ids)
            self.names = [(x["name"], x["bytes_per_pixel"]) for x in info]
        evs = []
        for i in range(len(self.names)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            hz.nrd.denoise_range(self.ids, i, 1)
            b.record()
            evs.append((a, b))
        self.events.append(evs)

    def pass_times_ms(self):
        import torch

        torch.cuda.synchronize()
        acc = {}
        for evs in self.events:
            for (name, bpp), (a, b) in zip(self.names, evs):
                t, n, _ = acc.get(name, (0.0, 0, bpp))
                acc[name] = (t + a.elapsed_time(b), n + 1, bpp)
        return {k: (t / n, bpp) for k, (t, n, bpp) in acc.items()}


# pass name -> prefix of the kernel name in the rocprofv3 counter tables (tools/summarize_profiles.py)
PASS_KERNEL = {
    "REBLUR::ClassifyTiles": "k_classify_tiles", "REBLUR::PrePass": "k_spatial<0", "REBLUR::Blur": "k_spatial<1",
    "REBLUR::PostBlur": "k_spatial<2", "REBLUR::TemporalAccumulation": "k_temporal_accumulation",
    "REBLUR::PrePassTemporalAccumulation": "k_prepass_temporal_accumulation", "REBLUR::PrepareInputs": "k_prepare_",
    "REBLUR::HistoryFix": "k_history_fix", "REBLUR::TemporalStabilization": "k_temporal_stabilization",
    "RELAX::ClassifyTiles": "k_classify_tiles", "RELAX::PrePass": "k_spatial<0", "RELAX::TemporalAccumulation": "k_temporal_accumulation",
    "RELAX::HistoryFix": "k_history_fix",
}


def _traffic_files(workload):
    import glob
    import re

    def build_order(path):  # r01v6 < r01v8 < r01v10 < r02v1: numeric, not alphabetical
        return [int(n) for n in re.findall(r"\d+", os.path.basename(path).split("_hbm_traffic_")[0])]

    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_hbm_traffic_%s.json" % workload)), key=build_order)


def traffic_source(workload, lib_path):
    """where roofline.traffic comes from: the committed counter table (newest for the workload), the library it was collected on and whether
    that is the library this run times (a lookup must not pass for a measurement of the build at hand)"""
    import hashlib
    import json

    files = _traffic_files(workload)
    if not files:
        return None
    on = json.load(open(files[-1])).get("_measured_on", {})
    mine = hashlib.sha256(open(lib_path, "rb").read()).hexdigest()[:12] if os.path.exists(lib_path) else None
    return {"table": "profiles/" + os.path.basename(files[-1]), "collected_with": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/profile_gpu.sh); not measured in this run",
            "table_library_sha256_12": on.get("library_sha256_12"), "timed_library_sha256_12": mine,
            "same_build": (on.get("library_sha256_12") == mine) if on.get("library_sha256_12") else None}


def measured_traffic(workload, pass_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x2 per the gfx950 correction of
    MI355X_MICROARCH.md + WRITE_SIZE, separate rocprofv3 --pmc runs, tools/profile_gpu.sh); None if that workload / kernel
    has no committed counter table - a live bench run cannot collect counters itself."""
    import glob
    import json

    prefix = PASS_KERNEL.get(pass_name)
    files = _traffic_files(workload)
    if not prefix or not files:
        return None
    table = json.load(open(files[-1]))
    for k, v in table.items():
        if k.startswith(prefix) and not k.startswith("_"):
            return round(v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"])
    return None


if __name__ == "__main__":
    main()
