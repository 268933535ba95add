"""Shared helpers of the parity tests."""
import numpy as np


def f16_ordered(a):
    """fp16 bit patterns -> integers ordered like the values (for ULP distances)."""
    i = np.ascontiguousarray(a).view(np.int16).astype(np.int32)
    return np.where(i < 0, -32768 - i, i)


def max_ulp_f16(a, b):
    return int(np.abs(f16_ordered(a) - f16_ordered(b)).max())


def psnr(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float("inf")
    peak = max(np.abs(b).max(), 1e-12)
    return 10.0 * np.log10(peak * peak / mse)


def default_settings(api, scene, denoisers, **reblur_kw):
    D = api.Denoiser
    s = {}
    for d in denoisers:
        if d.name.startswith("REBLUR"):
            s[d] = api.ReblurSettings(**reblur_kw)
        elif d in (D.SIGMA_SHADOW, D.SIGMA_SHADOW_TRANSLUCENCY):
            s[d] = api.SigmaSettings(lightDirection=list(scene.sun))
        elif d == D.REFERENCE:
            s[d] = api.ReferenceSettings()
        elif d.name.startswith("RELAX"):
            kw = {k: v for k, v in reblur_kw.items() if k in ("minMaterialForDiffuse", "minMaterialForSpecular")}
            s[d] = api.RelaxSettings(**kw)
    return s


def run_frames(api, harness_mod, backend, scene, denoisers, nframes, settings=None, common_hook=None, frame_hook=None, keep=None, threads=None):
    """Run `nframes` through a fresh Harness; returns the harness (outputs of the last frame stay bound). `threads`: row-stripe the
    CPU oracle over that many host threads (large frames)."""
    h = harness_mod.Harness(backend, denoisers, scene.w, scene.h)
    if threads and hasattr(backend.lib, "orc_set_threads"):
        backend.lib.orc_set_threads(h.nrd.handle, int(threads))
    settings = settings or default_settings(api, scene, denoisers)
    for f in range(nframes):
        fr = scene.frame(f)
        if frame_hook:
            frame_hook(f, fr)
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        if common_hook:
            common_hook(f, cs)
        h.frame(cs, h.upload(fr), settings)
        if keep is not None:
            keep.append({k: h.fetch(v).copy() for k, v in h.outputs.items()})
    return h


def host_frame(fr):
    """a frame rendered on the GPU (synth.Scene(device="cuda:0"): torch tensors) as the numpy planes the CPU oracle takes"""
    host = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in fr.items()}
    host["normal_roughness"] = host["normal_roughness"].view(np.uint32)
    return host


def compare_all(ha, hb, exact=True, ulp=1):
    """compare outputs and every pool plane of two harnesses; returns list of (name, detail) mismatches"""
    bad = []
    for key in ("out_diff", "out_spec", "out_diff_sh1", "out_spec_sh1", "out_diff_dirocc"):
        a, b = ha.fetch(ha.outputs[key]), hb.fetch(hb.outputs[key])
        if exact:
            if not np.array_equal(a, b):
                bad.append((key, int((a != b).sum())))
        else:
            u = max_ulp_f16(a.view(np.float16), b.view(np.float16))
            if u > ulp:
                bad.append((key, "%d ulp" % u))
    for key in ("out_diff_hitdist", "out_spec_hitdist"):  # R16_UNORM outputs of the OCCLUSION variants
        a, b = ha.fetch(ha.outputs[key]).view(np.uint16).astype(np.int32), hb.fetch(hb.outputs[key]).view(np.uint16).astype(np.int32)
        if (exact and not np.array_equal(a, b)) or np.abs(a - b).max() > 16:  # 1 ULP fp16 at 1.0 = 2^-11 = 32 LSB of unorm16
            bad.append((key, int(np.abs(a - b).max())))
    a, b = ha.fetch(ha.outputs["out_shadow"]).astype(np.int32), hb.fetch(hb.outputs["out_shadow"]).astype(np.int32)
    if (exact and not np.array_equal(a, b)) or np.abs(a - b).max() > 1:
        bad.append(("out_shadow", int(np.abs(a - b).max())))
    a, b = ha.fetch(ha.outputs["out_validation"]), hb.fetch(hb.outputs["out_validation"])
    if not np.array_equal(a, b):
        bad.append(("out_validation", int((a != b).sum())))
    if exact:
        for pool in (0, 1):
            for pa, pb in zip(ha.nrd.pools[pool], hb.nrd.pools[pool]):
                x, y = ha.fetch(pa["buf"]), hb.fetch(pb["buf"])
                if not np.array_equal(x, y):
                    bad.append((pa["name"], int((x != y).sum())))
    return bad


# ---- hand-built frames for the known-answer tests -----------------------------------------------------------------
def flat_frame(pkg, w, h, z=5.0, rough=0.5, diff_rgb=(0.6, 0.5, 0.4), spec_rgb=(0.3, 0.4, 0.5), norm_hit=0.5, rng=None, sigma=0.5):
    """camera-facing plane at constant view depth, zero motion; optional multiplicative noise on the radiance"""
    synth = pkg.synth
    n = np.zeros((h, w, 3))
    n[..., 2] = -1.0
    fr = {}
    fr["viewz"] = np.full((h, w), z, dtype=np.float32)
    fr["mv"] = np.zeros((h, w, 4), dtype=np.float16)
    fr["normal_roughness"] = synth.pack_normal_roughness(n, np.full((h, w), rough), np.zeros((h, w), dtype=np.uint32))
    for key, rgb in (("diff", diff_rgb), ("spec", spec_rgb)):
        c = np.broadcast_to(np.asarray(rgb, dtype=np.float64), (h, w, 3)).copy()
        if rng is not None:
            c *= np.exp(sigma * rng.standard_normal((h, w, 1)) - 0.5 * sigma * sigma)
        fr[key] = np.concatenate([synth.linear_to_ycocg(c), np.full((h, w, 1), norm_hit)], -1).astype(np.float16)
    fr["penumbra"] = np.full((h, w), 65504.0, dtype=np.float16)
    fr["translucency"] = np.zeros((h, w, 4), dtype=np.uint8)
    fr["translucency"][..., 0] = 255
    fr["confidence"] = np.ones(((h + 4) // 5, (w + 4) // 5, 4), dtype=np.float16)
    fr["signal"] = np.concatenate([np.broadcast_to(np.asarray(diff_rgb), (h, w, 3)), np.ones((h, w, 1))], -1).astype(np.float16)
    return fr


def static_common(api, w, h, frame_index=0, reset=False, res_w=None, res_h=None, denoising_range=100.0):
    cs = api.CommonSettings()
    aspect = w / h
    for m in (cs.viewToClipMatrix, cs.viewToClipMatrixPrev):
        m[0] = 1.0
        m[5] = aspect
        m[10] = 1.0
        m[11] = 1.0
        m[14] = -0.05
    for m in (cs.worldToViewMatrix, cs.worldToViewMatrixPrev):
        for i in (0, 5, 10, 15):
            m[i] = 1.0
    cs.motionVectorScale[0] = 1.0 / w
    cs.motionVectorScale[1] = 1.0 / h
    cs.motionVectorScale[2] = 1.0
    for k in ("resourceSize", "resourceSizePrev"):
        getattr(cs, k)[0] = res_w or w
        getattr(cs, k)[1] = res_h or h
    for k in ("rectSize", "rectSizePrev"):
        getattr(cs, k)[0] = w
        getattr(cs, k)[1] = h
    cs.denoisingRange = denoising_range
    cs.frameIndex = frame_index
    cs.accumulationMode = int(api.AccumulationMode.CLEAR_AND_RESTART if reset else api.AccumulationMode.CONTINUE)
    cs.isHistoryConfidenceAvailable = False
    return cs
