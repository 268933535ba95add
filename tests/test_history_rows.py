"""nrdhip_set_history_rows (include/nrdhip.h): the row tilers refresh only the rows of last frame's permanent planes that reprojection may
reach (owned rows +- motion allowance + 2), not the whole stored halo (round 4). A footprint beyond that window must be REJECTED - never
read: what sits in the un-exchanged halo rows is last-but-one frame's data at best. ADVICE r4 (medium): the kernels accepted any stored row."""
import numpy as np
import pytest

import util

W, H = 64, 96
WINDOW = (40, 24)  # local rows [40, 64) hold current previous-frame state


def run(pkg, api, backend, dens, poison, window, tilt):
    """frame 0 from a restart on a camera-facing plane (every footprint validates), then the permanent planes are poisoned outside `window`,
    then frame 1 with a screen-space motion of 6 rows: the footprints of the rows next to the window's lower edge leave it"""
    rng = np.random.default_rng(5)
    st = {d: (api.ReblurSettings() if d.name.startswith("REBLUR") else api.RelaxSettings() if d.name.startswith("RELAX") else api.SigmaSettings(lightDirection=[0.0, 0.0, -1.0]))
          for d in dens}
    hz = pkg.harness.Harness(backend, dens, W, H)
    if hasattr(backend.lib, "orc_set_threads"):
        backend.lib.orc_set_threads(hz.nrd.handle, 4)
    for f in range(2):
        fr = util.flat_frame(pkg, W, H, rng=rng)
        # SIGMA: half of the pixels lit (no occluder: FP16_MAX), half behind an occluder with a wide penumbra - a noisy soft shadow everywhere,
        # so that the stabilization's history clamp has a range
        fr["penumbra"] = np.where(rng.random((H, W)) < 0.5, 65504.0, 0.02 + 0.05 * rng.random((H, W))).astype(np.float16)
        if f == 1:
            fr["mv"][..., 1] = np.float16(6.0)
            if window:
                hz.nrd.set_history_rows(*window)
            if poison:
                lo, hi = WINDOW[0], WINDOW[0] + WINDOW[1]
                for p in hz.nrd.pools[0]:
                    if "Guide" in p["name"]:
                        continue  # the geometry of the stale rows stays plausible (it is what validates a footprint); their SIGNAL is the poison
                    buf = hz.fetch(p["buf"])
                    buf[:lo] = 0x5B  # finite garbage in every byte (0x5B5B as fp16 = 235.4, accumulation codes 0x5B)
                    buf[hi:] = 0x5B
        hz.frame(util.static_common(api, W, H, f, reset=(f == 0)), hz.upload(fr), st)
    keys = ("out_diff", "out_spec", "out_shadow")
    return {k: np.array(hz.fetch(hz.outputs[k])) for k in keys}, hz


@pytest.mark.parametrize("den", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR", "SIGMA_SHADOW_TRANSLUCENCY"])
def test_footprints_outside_the_history_window_are_rejected_not_read(pkg, api, oracle, emulated, den):
    dens = [api.Denoiser[den]]
    tilt = {}
    for backend in (oracle, emulated):
        clean, _ = run(pkg, api, backend, dens, poison=False, window=WINDOW, tilt=tilt)
        dirty, _ = run(pkg, api, backend, dens, poison=True, window=WINDOW, tilt=tilt)
        for k in clean:
            assert np.array_equal(clean[k], dirty[k]), "%s: %s depends on rows outside the history window" % (backend.prefix, k)
        # the window is what protects: without it the same poison reaches the output
        unprotected, _ = run(pkg, api, backend, dens, poison=True, window=None, tilt=tilt)
        assert any(not np.array_equal(clean[k], unprotected[k]) for k in clean)
    # both sides implement the same window (bit for bit)
    a, _ = run(pkg, api, oracle, dens, poison=False, window=WINDOW, tilt=tilt)
    b, _ = run(pkg, api, emulated, dens, poison=False, window=WINDOW, tilt=tilt)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    # and a window that covers every stored row changes nothing
    full, _ = run(pkg, api, oracle, dens, poison=False, window=(0, H), tilt=tilt)
    none, _ = run(pkg, api, oracle, dens, poison=False, window=None, tilt=tilt)
    for k in full:
        assert np.array_equal(full[k], none[k]), k
