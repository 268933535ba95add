"""Worker of tests/test_tiler_gloo.py: one rank of a row-tiled run on the CPU (gloo), oracle backend standing in for the kernels.
Launched by torch.distributed.run; writes the rows it owns to <outdir>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import __graft_entry__ as graft  # noqa: E402


def main():
    outdir, w, frame_h, nframes, halo = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    mode = sys.argv[6] if len(sys.argv) > 6 else "default"
    pkg = graft.load_package()
    api = pkg.api
    from nrd_sample_amd import tiler

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    use_hip = len(sys.argv) > 7 and sys.argv[7] == "hip"  # -m gpu variant: the real kernels on cuda:0 (all ranks share the one GPU of the box)
    use_emu = len(sys.argv) > 7 and sys.argv[7] == "emu"  # the kernel sources compiled for the host (tests/hip_emu): band offsets through the kernels on CPU
    if use_emu:
        orc = api.Backend(graft.build_emulated(), "nrdhip_", "cpu")
    else:
        orc = pkg.hip_backend("cuda:0") if use_hip else graft.oracle_backend()
    native = len(sys.argv) > 8 and sys.argv[8] == "native"  # the C++ row tiler below the C-ABI, rows moved by gloo through its transport callbacks
    sys.path.insert(0, HERE)
    import test_tiler_gloo as tt
    scene = pkg.synth.Scene(w, frame_h, dolly=0.03, denoiser="RELAX" if mode == "relax8" else "REBLUR")  # every rank renders the same global frame and keeps its window
    dens, st = tt.case_of(api, scene, mode)
    if halo <= 0:  # derive the stored halo from the settings (reach of every pass + motion)
        halo = tiler.probe_halo(orc, dens, st)
    band = tiler.BandHarness(orc, dens, w, frame_h, rank, world, halo=halo)
    t = tiler.NativeTiler(band, dist, transport="dist") if native else tiler.Tiler(band, dist)
    blob = {}
    for f in range(nframes):
        fr = scene.frame(f)
        tt.mode_frame_hook(pkg, mode, f, fr)
        local = {k: band.local_rows(v) for k, v in fr.items() if k in ("viewz", "mv", "normal_roughness", "diff", "spec", "penumbra", "translucency")}
        local["confidence"] = fr["confidence"]
        # poison the halo rows of the inputs: the tiler must refresh them from the neighbours
        L = band.layout
        for k, v in local.items():
            if k == "confidence":
                continue
            v = np.ascontiguousarray(v).copy()
            v[:L["own_first"]] = 0
            v[L["own_first"] + L["own_rows"]:] = 0
            local[k] = v
        planes = band.upload(local)
        t.exchange_inputs(planes)
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        band.nrd.new_frame()
        band.nrd.set_common_settings(cs)
        band.bind(planes)
        for d in dens:
            band.nrd.set_denoiser_settings(int(d), st[d])
        t.denoise([int(d) for d in dens])
        for key in ("out_diff", "out_spec", "out_shadow"):
            blob["f%d_%s" % (f, key)] = band.own_rows(band.fetch(band.outputs[key])).copy()
    t.finish()  # halo rows of the permanent planes written by the last frame are still travelling
    blob["history"] = band.own_rows(band.pool("RELAX::History" if mode == "relax8" else "REBLUR::History")).copy()
    blob["halo"] = np.array([halo])
    blob["own0"] = np.array([band.layout["own0"], band.layout["own1"]])
    blob["bytes"] = np.array([t.bytes_exchanged])
    blob["split"] = np.array([t.split_dispatches])
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **blob)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
