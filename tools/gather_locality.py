"""How well do the taps of a pass coalesce? Runs the UNMODIFIED kernel sources in the host emulation (tests/hip_emu, no GPU) on the
bench scene and counts, for every wave (64 consecutive threads of a workgroup: 16 x 4 pixels) and every buffer gather it executes, the
distinct 128-byte lines the 64 lanes touch (tests/hip_emu/hip/hip_runtime.h GatherTrace). 1 line = perfectly coalesced; a 16 x 4-texel
group of 8-byte texels that moves rigidly touches 4-8; 64 = every lane somewhere else.

    python tools/gather_locality.py [width height frames]      (default 1920 1080 10; a minute or two)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402


def main():
    w, h, frames = (int(a) for a in (sys.argv[1:4] + ["1920", "1080", "10"][len(sys.argv) - 1:]))
    pkg = graft.load_package()
    api = pkg.api
    import util

    emu = api.Backend(graft.build_emulated(), "nrdhip_", "cpu")  # (what the tests' `emulated` fixture loads)
    trace = emu.lib.nrdhip_debug_gather_trace
    trace.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    trace.restype = None
    D = api.Denoiser
    dens = [D.REBLUR_DIFFUSE_SPECULAR]
    scene = pkg.synth.Scene(w, h, dolly=0.01)
    st = util.default_settings(api, scene, dens, minMaterialForDiffuse=0, minMaterialForSpecular=1)
    hz = pkg.harness.Harness(emu, dens, w, h)
    for f in range(frames - 1):
        fr = scene.frame(f)
        hz.frame(scene.common_settings(api, fr, f, reset=(f == 0)), hz.upload(fr), st)
    f = frames - 1
    fr = scene.frame(f)
    hz.nrd.new_frame()
    hz.nrd.set_common_settings(scene.common_settings(api, fr, f))
    hz.bind(hz.upload(fr))
    hz.nrd.set_denoiser_settings(int(dens[0]), st[dens[0]])
    ids = [int(dens[0])]
    info = hz.nrd.dispatches(ids)
    out = (ctypes.c_double * 3)()
    print("%dx%d, frame %d of a dolly: distinct 128-byte lines per wave-level gather" % (w, h, f))
    for i, d in enumerate(info):
        trace(1, None)
        hz.nrd.denoise_range(ids, i, 1)
        trace(0, out)
        if out[0] > 0:
            print("  %-32s %5.1f lines per gather   (%4.1f gathers per pixel, %4.1f lanes active per gather)" % (
                d["name"], out[1] / out[0], out[2] / (w * h), out[2] / out[0]))
        else:
            print("  %-32s no buffer gathers (loads through plain pointers)" % d["name"])


if __name__ == "__main__":
    main()
