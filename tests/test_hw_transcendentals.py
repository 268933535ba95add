"""The NRD_HW_TRANSCENDENTALS build flavour (libnrdhip_hwt.so: v_rcp_f32 / v_sqrt_f32 / v_exp_f32 in the weight arithmetic of the spatial
filters, csrc/nrd_device.h) against its checker (liboracle_hwt.so: IEEE 1 / x, sqrtf, exp2f at the same places).

The instructions are 1 ULP, not correctly rounded, so this flavour cannot be bit-identical to any CPU statement. VERDICT r4 asked whether it
holds north_star's bar - every OUT_* plane and the history within 1 ULP fp16, PSNR >= 60 dB - over 36 frames at the bench's operating
point, where differences have the whole accumulation length to compound. MEASURED (round 5, profiles/r05_hwt_distance_*.json,
profiles/r05_ab_hw_transcendentals.txt): it does NOT. PSNR is 97-124 dB and 99.94 % (REBLUR) / 99.99 % (RELAX SH) of the values stay within
1 ULP, but isolated pixels drift by hundreds of ULP: weights that differ in their last bit move a few fp16 roundings, a moved signal moves an
accumulation-speed code now and then (1.6e-5 of the codes: decision-class arithmetic is exact on both sides, its INPUTS are not), and
a pixel with another code places its Blur taps elsewhere. For +3.4 % on the headline (+7 % without sky) the flavour therefore stays an
OPTION; libnrdhip.so remains the bit-reproducible build. This test holds the option to what it does deliver and prints the full distance
histogram (share of values that differ at all, by 1 ULP, by more; the largest distance) into the log of every GPU run."""
import json
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

FRAMES, FIRST_CHECKED = 36, 30


def ulp_hist(a16, b16):
    d = np.abs(util.f16_ordered(a16) - util.f16_ordered(b16)).ravel()
    n = d.size
    return {"values": int(n), "differ": float((d > 0).sum() / n), "ulp1": float((d == 1).sum() / n), "ulp2_4": float(((d >= 2) & (d <= 4)).sum() / n),
            "ulp5_16": float(((d >= 5) & (d <= 16)).sum() / n), "ulp_gt16": float((d > 16).sum() / n), "max_ulp": int(d.max())}


def run(pkg, api, oracle, hip, dens, w, h, threads, dolly=0.004):
    import bench

    scene = pkg.synth.Scene(w, h, dolly=dolly, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR", device="cuda:0")
    dd = [api.Denoiser[x] for x in dens]
    st = bench.settings_of(api, scene, dd)
    ho = pkg.harness.Harness(oracle, dd, w, h)
    oracle.lib.orc_set_threads(ho.nrd.handle, threads)
    hg = pkg.harness.Harness(hip, dd, w, h)
    rows = []
    hist_name = ("RELAX" if dens[0].startswith("RELAX") else "REBLUR") + "::History"
    for f in range(FRAMES):
        fr = scene.frame(f)
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        ho.frame(cs, ho.upload(util.host_frame(fr)), st)
        hg.frame(cs, hg.upload(fr), st)
        if f >= FIRST_CHECKED:
            for key in ("out_diff", "out_spec", "out_diff_sh1", "out_spec_sh1"):
                a, b = ho.fetch(ho.outputs[key]).view(np.float16), hg.fetch(hg.outputs[key]).view(np.float16)
                if not a.any():
                    continue
                r = ulp_hist(a, b)
                r.update(frame=f + 1, plane=key, psnr_db=round(util.psnr(b.astype(np.float32), a.astype(np.float32)), 2))
                rows.append(r)
            a, b = np.asarray(ho.pool(hist_name)).view(np.float16), np.asarray(hg.pool(hist_name)).view(np.float16)
            r = ulp_hist(a, b)
            r.update(frame=f + 1, plane=hist_name, psnr_db=round(util.psnr(b.astype(np.float32), a.astype(np.float32)), 2))
            rows.append(r)
    # accumulation speeds are decision-class (exact sequences on both sides): how many codes moved anyway, through the signals?
    codes = {}
    for name in [p["name"] for p in ho.nrd.pools[0] if "Data1" in p["name"] or "HistoryLength" in p["name"]]:
        a, b = np.asarray(ho.pool(name)), np.asarray(hg.pool(name))
        codes[name] = float((a != b).mean())
    return rows, codes


def record(tag, rows, codes):
    worst = max(r["max_ulp"] for r in rows)
    summary = {"flavour": "hwt vs liboracle_hwt", "case": tag, "frames_checked": "%d..%d" % (FIRST_CHECKED + 1, FRAMES), "max_ulp": worst,
               "min_psnr_db": min(r["psnr_db"] for r in rows), "max_differ_frac": max(r["differ"] for r in rows),
               "max_frac_gt1ulp": max(r["ulp2_4"] + r["ulp5_16"] + r["ulp_gt16"] for r in rows), "accum_code_mismatch_frac": codes, "rows": rows}
    print("HWT-DISTANCE " + json.dumps(summary))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "hwt_distance_%s.json" % tag), "w") as f:
            json.dump(summary, f, indent=1)
    return summary


def held_to(s):
    """what the optional flavour is held to (the product bar - max <= 1 ULP - is what it was measured NOT to meet: see the module docstring)"""
    print("strict bar (max <= 1 ULP fp16 everywhere) met: %s (max %d ULP)" % (s["max_ulp"] <= 1, s["max_ulp"]))
    assert s["min_psnr_db"] >= 90.0, s["min_psnr_db"]  # north_star asks for 60
    assert s["max_frac_gt1ulp"] <= 2e-3, s["max_frac_gt1ulp"]  # measured 5.8e-4 (REBLUR out_diff) / 9.7e-5 (RELAX SH1)
    assert s["max_differ_frac"] <= 3e-2, s["max_differ_frac"]  # measured 9.6e-3
    assert all(v <= 2e-4 for v in s["accum_code_mismatch_frac"].values()), s["accum_code_mismatch_frac"]  # measured 1.6e-5


def test_hwt_reblur_1080p_36_frames(pkg, api, oracle_hwt, hip_hwt):
    rows, codes = run(pkg, api, oracle_hwt, hip_hwt, ["REBLUR_DIFFUSE_SPECULAR"], 1920, 1080, 128)
    s = record("reblur_ds_1080p", rows, codes)
    held_to(s)


def test_hwt_relax_sh_720p_36_frames(pkg, api, oracle_hwt, hip_hwt):
    rows, codes = run(pkg, api, oracle_hwt, hip_hwt, ["RELAX_DIFFUSE_SPECULAR_SH"], 1280, 720, 128)
    s = record("relax_ds_sh_720p", rows, codes)
    held_to(s)
