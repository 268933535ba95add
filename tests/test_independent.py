"""an independent check of the oracle that shares no code with oracle/orc_math.h - numpy / float64 restatements
(tests/indep/reblur_numpy.py) of REFERENCE accumulation, the guide decode and one full REBLUR spatial pass (PrePass, both
signals) on the committed golden inputs (tests/golden/inputs_64x48.npz), held to <= 1 fp16 ULP; and the MEASUREMENT behind the
deviation ledger of oracle/README.md: how far each knowingly non-upstream formula moves that pass's output."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "indep"))
import reblur_numpy as ind  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "inputs_64x48.npz"))
W, H = 64, 48


def frame(f):
    return {k[len("f%d_" % f):]: GOLD[k] for k in GOLD.files if k.startswith("f%d_" % f)}


def ulp16(a, b):
    a = a.view(np.int16).astype(np.int32)
    b = b.view(np.int16).astype(np.int32)
    a = np.where(a < 0, -32768 - a, a)
    b = np.where(b < 0, -32768 - b, b)
    return np.abs(a - b)


def settings_dict(st):
    hp = st.hitDistanceParameters
    return dict(planeDistanceSensitivity=st.planeDistanceSensitivity, roughnessFraction=st.roughnessFraction, minHitDistanceWeight=st.minHitDistanceWeight,
                diffusePrepassBlurRadius=st.diffusePrepassBlurRadius, specularPrepassBlurRadius=st.specularPrepassBlurRadius,
                minMaterialForDiffuse=st.minMaterialForDiffuse, minMaterialForSpecular=st.minMaterialForSpecular, hitDistanceParameters=(hp.A, hp.B, hp.C, hp.D))


def oracle_prepass(pkg, api, oracle, f):
    """Tmp1 (both signals) and the tracked specular hit distance after ClassifyTiles + PrePass of frame f on a fresh instance"""
    D = api.Denoiser
    scene = pkg.synth.Scene(W, H, dolly=0.03)
    fr = frame(f)
    hz = pkg.harness.Harness(oracle, [D.REBLUR_DIFFUSE_SPECULAR], W, H, separate_passes=True)  # (the PrePass result as a plane: no fused dispatch)
    st = api.ReblurSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1)
    cs = scene.common_settings(api, fr, f, reset=True)
    hz.nrd.new_frame()
    hz.nrd.set_common_settings(cs)
    hz.bind(hz.upload(fr))
    hz.nrd.set_denoiser_settings(int(D.REBLUR_DIFFUSE_SPECULAR), st)
    names = [d["name"] for d in hz.nrd.dispatches([int(D.REBLUR_DIFFUSE_SPECULAR)])]
    assert names[:2] == ["REBLUR::ClassifyTiles", "REBLUR::PrePass"]
    hz.nrd.denoise_range([int(D.REBLUR_DIFFUSE_SPECULAR)], 0, 2)
    tmp1 = hz.pool("REBLUR::Tmp1").copy().view(np.float16).reshape(H, W, 2, 4)
    track = hz.pool("REBLUR::SpecHitDistForTracking").copy().view(np.float16).reshape(H, W)
    guide = hz.pool("REBLUR::Guide_A").copy() if cs.frameIndex % 2 == 0 else hz.pool("REBLUR::Guide_B").copy()
    return fr, cs, st, tmp1, track, guide


def test_reference_accumulation_independent(pkg, api, oracle):
    D = api.Denoiser
    scene = pkg.synth.Scene(W, H, dolly=0.03)
    hz = pkg.harness.Harness(oracle, [D.REFERENCE], W, H)
    hist = None
    for f in range(4):
        fr = frame(f)
        sig = fr["signal"].copy()
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        hz.frame(cs, hz.upload({"signal": sig}), {D.REFERENCE: api.ReferenceSettings()})
        hist, want = ind.reference_accumulate(hist, fr["signal"], f, api.ReferenceSettings().maxAccumulatedFrameNum, restart=(f == 0))
        out = hz.nrd._bound[int(api.ResourceType.OUT_SIGNAL)]
        assert ulp16(np.asarray(out).view(np.float16).reshape(H, W, 4), want).max() <= 1, f
        assert np.abs(hz.pool("REFERENCE::History").view(np.float32).reshape(H, W, 4) - hist).max() <= 2e-7 * max(1.0, float(np.abs(hist).max()))


@pytest.mark.parametrize("flavour", ["default", "frozen"])
@pytest.mark.parametrize("f", [0, 2])
def test_guide_and_prepass_independent(request, pkg, api, f, flavour):
    """both build flavours: the default (hit-distance weight exp(-3|x|), normal weight on the angle as upstream's AcosApprox takes it, the chord - the
    restatement takes exact exp / sqrt where the oracle evaluates a polynomial / a Newton iteration) and the frozen one ((1 - |x|)^2, squared angle)"""
    upstream = flavour == "default"
    fr, cs, st, tmp1, track, guide = oracle_prepass(pkg, api, request.getfixturevalue("oracle" if upstream else "oracle_frozen"), f)
    # guide texel {viewZ 22 bit | roughness code, normal 3 x 10 bit | materialID}: depth word and material exact; a normal code may sit
    # one step off where the float32 octahedral decode and the float64 one round to different sides (a handful of texels)
    w0, w1 = ind.guide_words(fr["viewz"], fr["normal_roughness"], denoising_range=cs.denoisingRange)
    g = guide.view(np.uint32).reshape(H, W, 2)
    assert np.array_equal(g[..., 0], w0) and np.array_equal(g[..., 1] >> 30, w1 >> 30)
    for sh in (0, 10, 20):
        d = np.abs(((g[..., 1] >> sh) & 1023).astype(np.int32) - ((w1 >> sh) & 1023).astype(np.int32))
        assert d.max() <= 1 and float((d == 0).mean()) > 0.98
    want, want_track = ind.prepass(fr["viewz"], fr["normal_roughness"], fr["diff"], fr["spec"], fr["view_to_clip"], fr["world_to_view"], cs.frameIndex, cs.denoisingRange,
                                   settings_dict(st), exp_hit_weight=upstream, angle_normal_weight=upstream)
    d = ulp16(tmp1, want)
    # float64 vs the oracle's float32: a tap position may floor to the neighbouring texel where the projected offset sits on a pixel
    # boundary - a different (equally valid) tap, not an arithmetic error; everything else must agree to 1 fp16 ULP
    frac_off = float((d > 1).any(axis=(2, 3)).mean())
    print("AGREE PrePass frame %d %s: pixels with a value beyond 1 ULP %.3f %%, values within 1 ULP %.4f %%, max %d ULP" % (f, flavour, 100 * frac_off, 100 * float((d <= 1).mean()), int(d.max())))
    assert frac_off == 0.0 and int(d.max()) <= 1, (frac_off, int(d.max()))  # stated bar (round 5): EVERY value within 1 fp16 ULP on the golden frames
    dt = ulp16(track, want_track)
    assert float((dt <= 1).mean()) > 0.98


def psnr(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    mse = ((a - b) ** 2).mean()
    return 99.0 if mse == 0 else 10.0 * np.log10(max(float(np.abs(b).max()), 1e-9) ** 2 / mse)


def test_deviation_ledger_measurements(pkg, api, oracle_frozen, capsys):
    """How far each knowingly non-upstream formula moves the PrePass output on the golden scene (frames 0 and 2): the numbers quoted
    in oracle/README.md 'deviation ledger'. Asserts only that they stay in the bands written there."""
    rows = {}
    for f in (0, 2):
        fr, cs, st, tmp1, track, _ = oracle_prepass(pkg, api, oracle_frozen, f)
        args = (fr["viewz"], fr["normal_roughness"], fr["diff"], fr["spec"], fr["view_to_clip"], fr["world_to_view"], cs.frameIndex, cs.denoisingRange, settings_dict(st))
        base, _ = ind.prepass(*args)
        for name in ("exp_hit_weight", "angle_normal_weight", "no_reach", "f32_guide"):
            alt, _ = ind.prepass(*args, **{name: True})
            d = ulp16(base, alt)
            rel = np.abs(alt.astype(np.float64) - base.astype(np.float64)) / np.maximum(np.abs(base.astype(np.float64)), 1e-3)
            r = rows.setdefault(name, dict(max_ulp=0, frac_changed=0.0, max_rel=0.0, psnr=99.0))
            r["max_ulp"] = max(r["max_ulp"], int(d.max()))
            r["frac_changed"] = max(r["frac_changed"], float((d > 1).mean()))
            r["max_rel"] = max(r["max_rel"], float(rel[..., :3].max()))
            r["psnr"] = min(r["psnr"], psnr(alt[..., :3], base[..., :3]))
    with capsys.disabled():
        for k, v in rows.items():
            print("deviation %-20s max %5d ULP fp16, %.1f %% of values move > 1 ULP, max relative %.3f, PSNR %.1f dB" % (k, v["max_ulp"], 100 * v["frac_changed"], v["max_rel"], v["psnr"]))
    # ---- rows 16-18: the DEFAULT flavour's cheap evaluations against the exact functions (round 4's A/B records carried
    # speed only). Base = the default flavour with exact sqrt / exp; each switch changes ONE evaluation.
    rows2 = {}
    for f in (0, 2):
        fr, cs, st, tmp1, track, _ = oracle_prepass(pkg, api, oracle_frozen, f)
        args = (fr["viewz"], fr["normal_roughness"], fr["diff"], fr["spec"], fr["view_to_clip"], fr["world_to_view"], cs.frameIndex, cs.denoisingRange, settings_dict(st))
        base, _ = ind.prepass(*args, exp_hit_weight=True, angle_normal_weight=True)
        for name in ("one_step_sqrt", "poly3_exp2", "arc_normal_weight"):
            alt, _ = ind.prepass(*args, exp_hit_weight=True, angle_normal_weight=True, **{name: True})
            d = ulp16(base, alt)
            rel = np.abs(alt.astype(np.float64) - base.astype(np.float64)) / np.maximum(np.abs(base.astype(np.float64)), 1e-3)
            r = rows2.setdefault(name, dict(max_ulp=0, differ=0.0, frac_changed=0.0, max_rel=0.0, psnr=199.0))
            r["max_ulp"] = max(r["max_ulp"], int(d.max()))
            r["differ"] = max(r["differ"], float((d > 0).mean()))
            r["frac_changed"] = max(r["frac_changed"], float((d > 1).mean()))
            r["max_rel"] = max(r["max_rel"], float(rel[..., :3].max()))
            r["psnr"] = min(r["psnr"], psnr(alt[..., :3], base[..., :3]) if (alt != base).any() else 199.0)
    with capsys.disabled():
        for k, v in rows2.items():
            print("default-flavour evaluation %-18s max %4d ULP fp16, %.3f %% of values differ, %.4f %% by > 1 ULP, max relative %.5f, PSNR %.1f dB" % (
                k, v["max_ulp"], 100 * v["differ"], 100 * v["frac_changed"], v["max_rel"], v["psnr"]))
    # the cheap evaluations stay far inside what "1 ULP fp16 / 60 dB" tolerates; the chord-for-arc substitution (upstream's own) is the big one
    # (measured: 1 ULP / 0.16 % differ / 122 dB; 1 ULP / 0.60 % / 107 dB; 5 ULP / 0.15 % beyond 1 ULP / 113 dB - oracle/README.md rows 16-18)
    assert rows2["one_step_sqrt"]["max_ulp"] <= 1 and rows2["one_step_sqrt"]["differ"] < 0.004 and rows2["one_step_sqrt"]["psnr"] > 115.0
    assert rows2["poly3_exp2"]["max_ulp"] <= 1 and rows2["poly3_exp2"]["differ"] < 0.012 and rows2["poly3_exp2"]["psnr"] > 100.0
    assert rows2["arc_normal_weight"]["max_ulp"] <= 8 and rows2["arc_normal_weight"]["frac_changed"] < 0.004 and rows2["arc_normal_weight"]["psnr"] > 105.0
    # the 8-byte guide (22-bit depth, 3 x 10-bit normal): the 1e-3 normal step tilts the tap basis enough to move ~3 % of the taps (30-pixel
    # radius) onto the neighbouring texel - a different, equally valid sample of a noisy input, not a weight error (the depth alone: 65 dB)
    assert rows["f32_guide"]["psnr"] > 45.0 and rows["f32_guide"]["frac_changed"] < 0.06
    assert rows["no_reach"]["frac_changed"] < 0.05    # the hard reach only bites on the longest taps at grazing angles
    assert rows["exp_hit_weight"]["psnr"] > 25.0 and rows["angle_normal_weight"]["psnr"] > 25.0  # weight-shape changes: visible, bounded


# ---- the temporal half: TemporalAccumulation, HistoryFix, TemporalStabilization, restated in numpy / float64 with
# no code shared with oracle/orc_math.h (tests/indep/reblur_temporal_numpy.py). Every pass gets the planes the oracle's own pass got
# (pool snapshots between dispatches), so a disagreement is that pass's and nobody else's; frames 1-3 of the golden inputs, each with
# the history the frames before it built.
import reblur_temporal_numpy as tmp  # noqa: E402


def temporal_settings(st):
    return dict(maxAccumulatedFrameNum=st.maxAccumulatedFrameNum, maxFastAccumulatedFrameNum=st.maxFastAccumulatedFrameNum,
                maxStabilizedFrameNum=st.maxStabilizedFrameNum, minMaterialForDiffuse=st.minMaterialForDiffuse,
                minMaterialForSpecular=st.minMaterialForSpecular, roughnessFraction=st.roughnessFraction, lobeAngleFraction=st.lobeAngleFraction,
                planeDistanceSensitivity=st.planeDistanceSensitivity, historyFixFrameNum=st.historyFixFrameNum,
                historyFixBasePixelStride=st.historyFixBasePixelStride, fastHistoryClampingSigmaScale=st.fastHistoryClampingSigmaScale,
                responsiveRoughnessThreshold=st.responsiveAccumulationSettings.roughnessThreshold,
                responsiveMinAccum=float(st.responsiveAccumulationSettings.minAccumulatedFrameNum),
                antilagSigmaScale=st.antilagSettings.luminanceSigmaScale, antilagSensitivity=st.antilagSettings.luminanceSensitivity)


def agree(name, got, want, min_frac, mask=None, max_ulp=None):
    """the stated bar of an independent restatement ("a stated max ULP + fraction", not ">= 99 %"): at least `min_frac` of
    the values within 1 fp16 ULP and, where `max_ulp` is given, NO value farther than that"""
    d = ulp16(got, want)
    if mask is not None:
        d = d[mask]
    frac = float((d <= 1).mean())
    print("AGREE %-40s within 1 ULP %.4f %%, within 2 ULP %.4f %%, max %d ULP" % (name, 100 * frac, 100 * float((d <= 2).mean()), int(d.max())))
    assert frac >= min_frac, "%s: only %.2f %% of the values within 1 fp16 ULP (max %d)" % (name, 100 * frac, int(d.max()))
    assert max_ulp is None or int(d.max()) <= max_ulp, "%s: max %d ULP fp16 (bar %d)" % (name, int(d.max()), max_ulp)
    return frac


def agree_texel(name, got, want, min_frac, scale_from=None):
    """the bar for texels whose components have BOTH signs (YCoCg-coded golden inputs fed to RELAX as linear RGB, SH1 = direction x luminance):
    weighted sums of such components cancel, so a small component can sit several of ITS OWN ULPs off while the texel is right to 1e-4.
    At least `min_frac` of the values within 1 fp16 ULP and EVERY value within 1 ULP of its texel's largest component (`scale_from`: of the
    largest component of THAT texel as well - an SH1 texel is a sum of direction x luminance terms whose directions cancel too, so its
    scale is the luminance of the signal's first texel)"""
    frac = agree(name, got, want, min_frac)
    scale = np.maximum(np.abs(want.astype(np.float64)).max(-1, keepdims=True), 2.0 ** -14)
    if scale_from is not None:
        scale = np.maximum(scale, np.abs(scale_from.astype(np.float64)).max(-1, keepdims=True))
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert (err <= 2.0 ** (np.floor(np.log2(scale)) - 10)).all(), "%s: a value farther than 1 ULP of its texel's largest component" % name
    return frac


@pytest.mark.parametrize("f,flavour", [(1, "frozen"), (2, "frozen"), (3, "frozen"), (2, "default"), (3, "default")])
def test_temporal_passes_independent(request, pkg, api, f, flavour):
    """every pass of the REBLUR_DIFFUSE_SPECULAR frame behind the PrePass against its numpy / float64 restatement, each fed the planes the
    oracle fed its own pass - in BOTH build flavours: the weights of HistoryFix's reconstruction, Blur and PostBlur are the ones that differ
    (normal weight on the chord, hit-distance weight exp(-3|x|), Blur rotation per pixel in the default build)"""
    upstream = flavour == "default"
    oracle_frozen = request.getfixturevalue("oracle" if upstream else "oracle_frozen")
    D = api.Denoiser
    den = int(D.REBLUR_DIFFUSE_SPECULAR)
    scene = pkg.synth.Scene(W, H, dolly=0.03)
    hz = pkg.harness.Harness(oracle_frozen, [D.REBLUR_DIFFUSE_SPECULAR], W, H, separate_passes=True)  # (pool snapshots between all seven passes)
    st = api.ReblurSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1)
    s = temporal_settings(st)
    for g in range(f):  # the history this frame inherits
        fr = frame(g)
        hz.frame(scene.common_settings(api, fr, g, reset=(g == 0)), hz.upload(fr), {D.REBLUR_DIFFUSE_SPECULAR: st})
    fr, prev = frame(f), frame(f - 1)
    cs = scene.common_settings(api, fr, f)
    hz.nrd.new_frame()
    hz.nrd.set_common_settings(cs)
    hz.bind(hz.upload(fr))
    hz.nrd.set_denoiser_settings(den, st)
    names = [d["name"].split("::")[1] for d in hz.nrd.dispatches([den])]
    assert names == ["ClassifyTiles", "PrePass", "TemporalAccumulation", "HistoryFix", "Blur", "PostBlur", "TemporalStabilization"]
    cur, old = ("_A", "_B") if f % 2 == 0 else ("_B", "_A")
    rad = lambda name: hz.pool(name).copy().view(np.float16).reshape(H, W, 2, 4)
    lum = lambda name: hz.pool(name).copy().view(np.float16).reshape(H, W, 2)
    c = tmp.Consts(fr, W, H, cs.denoisingRange, cs.disocclusionThreshold)
    gcur = ind.decode_guide(fr["viewz"], fr["normal_roughness"])
    gprev = ind.decode_guide(prev["viewz"], prev["normal_roughness"])
    geo = np.abs(gcur[0]) <= cs.denoisingRange

    # ---- TemporalAccumulation
    hz.nrd.denoise_range([den], 0, 2)
    tmp1, hist, fast_prev = rad("REBLUR::Tmp1"), rad("REBLUR::History"), lum("REBLUR::FastHistory" + old)
    speeds_prev = hz.pool("REBLUR::Data1" + old).copy().view(np.uint16).reshape(H, W)
    track = hz.pool("REBLUR::SpecHitDistForTracking").copy().view(np.float16).reshape(H, W)
    hz.nrd.denoise_range([den], 2, 1)
    tmp2, fast, speeds_tmp = rad("REBLUR::Tmp2"), lum("REBLUR::FastHistory" + cur), hz.pool("REBLUR::Data1_Tmp").copy().view(np.uint16).reshape(H, W)
    data2 = hz.pool("REBLUR::Data2").copy().view(np.uint32).reshape(H, W)
    w_tmp2, w_fast, w_speeds, w_data2, info = tmp.temporal_accumulation(c, s, gcur, gprev, fr["mv"], tmp1, hist, fast_prev, speeds_prev, track, fr["confidence"], True)
    assert info["smb_ok"][geo].mean() > 0.8 and info["vmb_ok"][geo].mean() > 0.5  # the test exercises both footprints, not the fallbacks
    agree("TA radiance", tmp2, w_tmp2, 1.0, max_ulp=1)
    agree("TA fast history", fast, w_fast, 1.0, max_ulp=1)
    # validity bits of the 2 x 4 footprint texels: the surface-motion ones must agree; the virtual-motion footprint tests texels against
    # the SURFACE's plane, which puts whole rows of them next to the threshold (the ground / wall seam of this scene) where float32 and
    # float64 decide differently for a percent of the pixels
    assert float(((data2 & 15) == (w_data2 & 15)).mean()) > 0.995
    assert float((((data2 >> 4) & 15) == ((w_data2 >> 4) & 15)).mean()) > 0.97
    assert float((np.abs(((data2 >> 8) & 255).astype(np.int32) - ((w_data2 >> 8) & 255).astype(np.int32)) <= 1).mean()) > 0.99  # virtual-motion amount
    for shift in (0, 8):  # accumulation speeds, quarter frames
        a, b = (speeds_tmp >> shift) & 255, (w_speeds >> shift) & 255
        assert float((np.abs(a.astype(np.int32) - b.astype(np.int32)) <= 1).mean()) > 0.99 and float((a == b).mean()) > 0.95

    # ---- HistoryFix (fed with the ORACLE's TemporalAccumulation outputs)
    hz.nrd.denoise_range([den], 3, 1)
    taps = [hz.pool("REBLUR::Tap_%s_A" % k).copy().view(np.uint32).reshape(H, W, 4) for k in ("Diff", "Spec")]
    speeds_cur = hz.pool("REBLUR::Data1" + cur).copy().view(np.uint16).reshape(H, W)
    w_sig, w_speeds_cur, (w0, w1) = tmp.history_fix(c, s, gcur, tmp2, speeds_tmp, fast, fr["viewz"], fr["normal_roughness"], upstream=upstream)
    for k in range(2):
        got = np.ascontiguousarray(taps[k][..., 2:4]).view(np.float16).reshape(H, W, 4)
        agree("HistoryFix signal %d" % k, got, w_sig[:, :, k], 1.0, max_ulp=1)
        assert np.array_equal(taps[k][..., 0], w0), "guide part of the tap texels: depth | roughness word"
        assert float((taps[k][..., 1] == w1).mean()) > 0.98 and np.array_equal(taps[k][..., 1] >> 30, w1 >> 30), "guide part of the tap texels: normal | material word"
    for shift in (0, 8):
        a, b = (speeds_cur >> shift) & 255, (w_speeds_cur >> shift) & 255
        assert float((np.abs(a.astype(np.int32) - b.astype(np.int32)) <= 1).mean()) > 0.99

    # ---- Blur and PostBlur (fed with the ORACLE's tap texels: HistoryFix's, then Blur's)
    sb = dict(settings_dict(st), maxBlurRadius=st.maxBlurRadius, minBlurRadius=st.minBlurRadius, lobeAngleFraction=st.lobeAngleFraction)
    tap_signal = lambda planes: np.stack([np.ascontiguousarray(t[..., 2:4]).view(np.float16).reshape(H, W, 4) for t in planes], 2)
    hz.nrd.denoise_range([den], 4, 1)
    taps_b = [hz.pool("REBLUR::Tap_%s_B" % k).copy().view(np.uint32).reshape(H, W, 4) for k in ("Diff", "Spec")]
    w_blur = ind.blur_pass(False, fr["viewz"], fr["normal_roughness"], tap_signal(taps), speeds_cur, fr["view_to_clip"], fr["world_to_view"], cs.frameIndex,
                           cs.denoisingRange, sb, upstream=upstream)
    agree("Blur", tap_signal(taps_b), w_blur, 1.0, mask=np.broadcast_to(geo[..., None, None], (H, W, 2, 4)), max_ulp=1)
    for k in range(2):  # the guide part travels through Blur untouched
        assert np.array_equal(taps_b[k][..., :2], taps[k][..., :2])
    hz.nrd.denoise_range([den], 5, 1)
    w_post = ind.blur_pass(True, fr["viewz"], fr["normal_roughness"], tap_signal(taps_b), speeds_cur, fr["view_to_clip"], fr["world_to_view"], cs.frameIndex,
                           cs.denoisingRange, sb, upstream=upstream)
    agree("PostBlur", rad("REBLUR::History"), w_post, 1.0, mask=np.broadcast_to(geo[..., None, None], (H, W, 2, 4)), max_ulp=1)

    # ---- TemporalStabilization (fed with the ORACLE's PostBlur output)
    post, stab_prev = rad("REBLUR::History"), lum("REBLUR::StabilizedLuma" + old)
    hz.nrd.denoise_range([den], 6, 1)
    stab = lum("REBLUR::StabilizedLuma" + cur)
    out = np.stack([hz.output("out_diff"), hz.output("out_spec")], 2)
    w_out, w_stab = tmp.temporal_stabilization(c, s, gcur, fr["mv"], post, speeds_cur, data2, stab_prev, track, True)
    # (measured 99.80-99.88 % / 99.74-99.84 %: the remainder are pixels whose virtual-motion footprint validates differently in float32
    # and float64 - its texels are tested against the SURFACE's plane, which parks rows of them on the threshold; a different footprint is
    # another history sample, hence hundreds of ULP at those pixels and none in between)
    agree("TS output", out, w_out, 0.997)
    agree("TS stabilized luma", stab, w_stab, 0.996)


@pytest.mark.parametrize("f,flavour", [(1, "frozen"), (3, "frozen"), (1, "default"), (3, "default")])
def test_relax_atrous_iterations_independent(request, pkg, api, f, flavour):
    """one variance-guided A-trous iteration of RELAX_DIFFUSE_SPECULAR, twice: iteration 0 (variance from the accumulated moments +
    the 3x3 spatial estimate of short histories, stride 1) and iteration 1 (stride 2, variance carried in the texel) - both build flavours
    (default: linear RGB texels, Rec.709 luminance, exp(-3 x) luminance weight, normal weight on the chord)"""
    upstream = flavour == "default"
    oracle_frozen = request.getfixturevalue("oracle" if upstream else "oracle_frozen")
    D = api.Denoiser
    den = int(D.RELAX_DIFFUSE_SPECULAR)
    scene = pkg.synth.Scene(W, H, dolly=0.03)
    hz = pkg.harness.Harness(oracle_frozen, [D.RELAX_DIFFUSE_SPECULAR], W, H)
    st = api.RelaxSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1)
    s = {k: getattr(st, k) for k in ("depthThreshold", "spatialVarianceEstimationHistoryThreshold", "specularVarianceBoost", "diffusePhiLuminance",
                                     "specularPhiLuminance", "diffuseMinLuminanceWeight", "specularMinLuminanceWeight", "lobeAngleFraction",
                                     "specularLobeAngleSlack", "luminanceEdgeStoppingRelaxation", "normalEdgeStoppingRelaxation",
                                     "roughnessEdgeStoppingRelaxation", "roughnessFraction", "enableRoughnessEdgeStopping", "minMaterialForDiffuse",
                                     "minMaterialForSpecular")}
    for g in range(f):
        fr = frame(g)
        hz.frame(scene.common_settings(api, fr, g, reset=(g == 0)), hz.upload(fr), {D.RELAX_DIFFUSE_SPECULAR: st})
    fr = frame(f)
    cs = scene.common_settings(api, fr, f)
    hz.nrd.new_frame()
    hz.nrd.set_common_settings(cs)
    hz.bind(hz.upload(fr))
    hz.nrd.set_denoiser_settings(den, st)
    names = [d["name"].split("::")[1] for d in hz.nrd.dispatches([den])]
    assert names[:6] == ["ClassifyTiles", "PrePass", "TemporalAccumulation", "HistoryFix", "Atrous0", "Atrous1"]
    cur = "_A" if f % 2 == 0 else "_B"
    c = tmp.Consts(fr, W, H, cs.denoisingRange, cs.disocclusionThreshold)
    gcur = ind.decode_guide(fr["viewz"], fr["normal_roughness"])
    rad = lambda name: hz.pool(name).copy().view(np.float16).reshape(H, W, 2, 4)
    hz.nrd.denoise_range([den], 0, 4)
    hist, moments = rad("RELAX::History"), hz.pool("RELAX::Moments" + cur).copy().view(np.float16).reshape(H, W, 2)
    speeds = hz.pool("RELAX::HistoryLength" + cur).copy().view(np.uint16).reshape(H, W)
    data2 = hz.pool("RELAX::Data2").copy().view(np.uint32).reshape(H, W)
    hz.nrd.denoise_range([den], 4, 1)
    a0 = rad("RELAX::Atrous_A")
    agree("A-trous iteration 0", a0, tmp.atrous_iteration(c, s, gcur, hist, 0, speeds, moments, data2, upstream=upstream), 1.0, max_ulp=1)
    hz.nrd.denoise_range([den], 5, 1)
    agree("A-trous iteration 1", rad("RELAX::Atrous_B"), tmp.atrous_iteration(c, s, gcur, a0, 1, data2=data2, upstream=upstream), 1.0, max_ulp=1)


@pytest.mark.parametrize("f,flavour", [(0, "default"), (2, "default"), (2, "frozen")])
def test_relax_prepass_independent(request, pkg, api, f, flavour):
    """RELAX_DIFFUSE_SPECULAR's PrePass (round 6): the spatial pass of tests/indep/reblur_numpy.py on RELAX's input convention - linear RGB
    (converted to YCoCg on the way in by the frozen flavour only) + world-space hit distances compared relative to the centre's"""
    upstream = flavour == "default"
    orc = request.getfixturevalue("oracle" if upstream else "oracle_frozen")
    D = api.Denoiser
    den = int(D.RELAX_DIFFUSE_SPECULAR)
    scene = pkg.synth.Scene(W, H, dolly=0.03)
    fr = frame(f)
    hz = pkg.harness.Harness(orc, [D.RELAX_DIFFUSE_SPECULAR], W, H)
    st = api.RelaxSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1)
    cs = scene.common_settings(api, fr, f, reset=True)
    hz.nrd.new_frame()
    hz.nrd.set_common_settings(cs)
    hz.bind(hz.upload(fr))
    hz.nrd.set_denoiser_settings(den, st)
    names = [d["name"] for d in hz.nrd.dispatches([den])]
    assert names[:2] == ["RELAX::ClassifyTiles", "RELAX::PrePass"]
    hz.nrd.denoise_range([den], 0, 2)
    tmp1 = hz.pool("RELAX::Tmp1").copy().view(np.float16).reshape(H, W, 2, 4)
    track = hz.pool("RELAX::SpecHitDistForTracking").copy().view(np.float16).reshape(H, W)
    s = dict(planeDistanceSensitivity=api.ReblurSettings().planeDistanceSensitivity, roughnessFraction=st.roughnessFraction,
             minHitDistanceWeight=st.minHitDistanceWeight, diffusePrepassBlurRadius=st.diffusePrepassBlurRadius,
             specularPrepassBlurRadius=st.specularPrepassBlurRadius, minMaterialForDiffuse=st.minMaterialForDiffuse,
             minMaterialForSpecular=st.minMaterialForSpecular, hitDistanceParameters=(1.0, 0.0, 1.0, 0.0))
    want, want_track = ind.prepass(fr["viewz"], fr["normal_roughness"], fr["diff"], fr["spec"], fr["view_to_clip"], fr["world_to_view"], cs.frameIndex,
                                   cs.denoisingRange, s, exp_hit_weight=upstream, angle_normal_weight=upstream, relax_in=True, to_ycocg=not upstream)
    # The golden inputs are YCoCg-coded texels fed to RELAX as if they were linear RGB: "colour" channels of both signs whose weighted
    # sums cancel (0.005 next to a luminance of 1.8), so a channel can sit 61 of ITS OWN ULPs off while the texel is right to 1e-4. Stated
    # bar: >= 99.99 % of the values within 1 fp16 ULP and EVERY value within 1 ULP of its texel's largest component
    agree_texel("RELAX PrePass frame %d %s" % (f, flavour), tmp1, want, 0.9999)
    assert float((ulp16(track, want_track) <= 1).mean()) > 0.98


@pytest.mark.parametrize("f,flavour", [(1, "default"), (2, "default"), (3, "default"), (2, "frozen")])
def test_relax_temporal_passes_independent(request, pkg, api, f, flavour):
    """RELAX_DIFFUSE_SPECULAR's TemporalAccumulation and HistoryFix (round 6) against the numpy / float64 restatement,
    each fed the planes the oracle fed its own pass: the two footprints with per-signal history caps, the fast and second-moment histories
    of the texel's LUMINANCE (Rec.709 of linear RGB in the default flavour, channel 0 of YCoCg in the frozen one), the reprojection quality
    the A-trous iterations relax their edge stopping by (data2 bits 16..23); HistoryFix: reconstruction with pow(N.Ns, power) as its normal
    weight, fast-history clamp of the luminance, antilag (acceleration + spatial / temporal sigma reset). Stated bars as for REBLUR"""
    upstream = flavour == "default"
    orc = request.getfixturevalue("oracle" if upstream else "oracle_frozen")
    D = api.Denoiser
    den = int(D.RELAX_DIFFUSE_SPECULAR)
    scene = pkg.synth.Scene(W, H, dolly=0.03)
    hz = pkg.harness.Harness(orc, [D.RELAX_DIFFUSE_SPECULAR], W, H)
    st = api.RelaxSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1)
    rb = api.ReblurSettings()  # what the shared passes take from ReblurSettings' defaults (oracle/orc_reblur.cpp relax_as_reblur)
    s = dict(maxAccumulatedFrameNum=st.diffuseMaxAccumulatedFrameNum, maxFastAccumulatedFrameNum=st.diffuseMaxFastAccumulatedFrameNum,
             minMaterialForDiffuse=st.minMaterialForDiffuse, minMaterialForSpecular=st.minMaterialForSpecular, roughnessFraction=st.roughnessFraction,
             lobeAngleFraction=st.lobeAngleFraction, planeDistanceSensitivity=rb.planeDistanceSensitivity, historyFixFrameNum=st.historyFixFrameNum,
             historyFixBasePixelStride=st.historyFixBasePixelStride, fastHistoryClampingSigmaScale=st.fastHistoryClampingSigmaScale,
             responsiveRoughnessThreshold=rb.responsiveAccumulationSettings.roughnessThreshold,
             responsiveMinAccum=float(rb.responsiveAccumulationSettings.minAccumulatedFrameNum))
    for g in range(f):
        fr = frame(g)
        hz.frame(scene.common_settings(api, fr, g, reset=(g == 0)), hz.upload(fr), {D.RELAX_DIFFUSE_SPECULAR: st})
    fr, prev = frame(f), frame(f - 1)
    cs = scene.common_settings(api, fr, f)
    hz.nrd.new_frame()
    hz.nrd.set_common_settings(cs)
    hz.bind(hz.upload(fr))
    hz.nrd.set_denoiser_settings(den, st)
    names = [d["name"].split("::")[1] for d in hz.nrd.dispatches([den])]
    assert names[:4] == ["ClassifyTiles", "PrePass", "TemporalAccumulation", "HistoryFix"]
    cur, old = ("_A", "_B") if f % 2 == 0 else ("_B", "_A")
    rad = lambda name: hz.pool(name).copy().view(np.float16).reshape(H, W, 2, 4)
    lum = lambda name: hz.pool(name).copy().view(np.float16).reshape(H, W, 2)
    u16 = lambda name: hz.pool(name).copy().view(np.uint16).reshape(H, W)
    c = tmp.Consts(fr, W, H, cs.denoisingRange, cs.disocclusionThreshold)
    gcur = ind.decode_guide(fr["viewz"], fr["normal_roughness"])
    gprev = ind.decode_guide(prev["viewz"], prev["normal_roughness"])
    geo = np.abs(gcur[0]) <= cs.denoisingRange

    # ---- TemporalAccumulation
    hz.nrd.denoise_range([den], 0, 2)
    tmp1, hist, fast_prev, mom_prev = rad("RELAX::Tmp1"), rad("RELAX::History"), lum("RELAX::FastHistory" + old), lum("RELAX::Moments" + old)
    speeds_prev = u16("RELAX::HistoryLength" + old)
    track = hz.pool("RELAX::SpecHitDistForTracking").copy().view(np.float16).reshape(H, W)
    hz.nrd.denoise_range([den], 2, 1)
    tmp2, fast, mom, speeds_tmp = rad("RELAX::Tmp2"), lum("RELAX::FastHistory" + cur), lum("RELAX::Moments" + cur), u16("RELAX::HistoryLength_Tmp")
    data2 = hz.pool("RELAX::Data2").copy().view(np.uint32).reshape(H, W)
    rx = dict(moments_prev=mom_prev, max_a_spec=st.specularMaxAccumulatedFrameNum, max_fast_spec=st.specularMaxFastAccumulatedFrameNum, rec709=upstream)
    w_tmp2, w_fast, w_speeds, w_data2, info = tmp.temporal_accumulation(c, s, gcur, gprev, fr["mv"], tmp1, hist, fast_prev, speeds_prev, track, fr["confidence"], True, relax=rx)
    assert info["smb_ok"][geo].mean() > 0.8 and info["vmb_ok"][geo].mean() > 0.5
    agree("RELAX TA radiance", tmp2, w_tmp2, 1.0, max_ulp=1)
    agree("RELAX TA fast history", fast, w_fast, 1.0, max_ulp=1)
    # (the second moment of this NOISY input spans five decades between neighbouring texels - 1e-5 next to 3.65 on frame 1: where float32 puts a
    # footprint 1e-6 of a texel off the row float64 puts it exactly on, the outlier below leaks in with that weight: 2 of 6144 values, 3 and 12 ULP)
    agree("RELAX TA second moment", mom, info["moments"], 0.999)
    assert float(((data2 & 15) == (w_data2 & 15)).mean()) > 0.995
    assert float((((data2 >> 4) & 15) == ((w_data2 >> 4) & 15)).mean()) > 0.97
    for shift in (8, 16):  # virtual-motion amount, reprojection quality of the specular history (unorm8)
        assert float((np.abs(((data2 >> shift) & 255).astype(np.int32) - ((w_data2 >> shift) & 255).astype(np.int32)) <= 1).mean()) > 0.99, shift
    for shift in (0, 8):  # accumulation speeds, quarter frames
        a, b = (speeds_tmp >> shift) & 255, (w_speeds >> shift) & 255
        assert float((np.abs(a.astype(np.int32) - b.astype(np.int32)) <= 1).mean()) > 0.99 and float((a == b).mean()) > 0.95

    # ---- HistoryFix + fast-history clamp + antilag (fed with the ORACLE's TemporalAccumulation outputs); the result IS the next history
    hz.nrd.denoise_range([den], 3, 1)
    new_hist, speeds_cur = rad("RELAX::History"), u16("RELAX::HistoryLength" + cur)
    al = st.antilagSettings
    rh = dict(moments=mom, normal_power=st.historyFixEdgeStoppingNormalPower, accel=al.accelerationAmount, spatial=al.spatialSigmaScale,
              temporal=al.temporalSigmaScale, reset=al.resetAmount, max_fast_spec=st.specularMaxFastAccumulatedFrameNum, rec709=upstream)
    w_sig, w_speeds_cur, _ = tmp.history_fix(c, s, gcur, tmp2, speeds_tmp, fast, fr["viewz"], fr["normal_roughness"], upstream=upstream, relax=rh)
    agree("RELAX HistoryFix history", new_hist, w_sig, 1.0, max_ulp=1)
    for shift in (0, 8):
        a, b = (speeds_cur >> shift) & 255, (w_speeds_cur >> shift) & 255
        assert float((np.abs(a.astype(np.int32) - b.astype(np.int32)) <= 1).mean()) > 0.99, shift
    # (the antilag must have had something to do on these frames: some pixel's history was shortened by the clamp / reset)
    assert ((speeds_cur & 255) < (speeds_tmp & 255)).any() or ((speeds_cur >> 8) < (speeds_tmp >> 8)).any()


def with_sh1(fr):
    """the golden frame plus SH1 planes for the SH denoisers (REBLUR_/RELAX_FrontEnd_PackSh, Shaders/TraceOpaque.cs.hlsl:738-752: direction x
    luminance of the sample) - built here from the frame's own normals so that oracle and restatement see the same bytes"""
    fr = dict(fr)
    n = ind.decode_guide(fr["viewz"], fr["normal_roughness"])[1]
    for key in ("diff", "spec"):
        rad = np.asarray(fr[key]).astype(np.float64)
        lum = 0.2126 * rad[..., 0] + 0.7152 * rad[..., 1] + 0.0722 * rad[..., 2]
        fr[key + "_sh1"] = np.concatenate([n * lum[..., None], np.zeros(lum.shape + (1,))], -1).astype(np.float16)
    return fr


@pytest.mark.parametrize("f", [1, 3])
def test_relax_sh_passes_independent(pkg, api, oracle, f):
    """RELAX_DIFFUSE_SPECULAR_SH - BASELINE config 4's denoiser - pass by pass (round 6): in every pass the second texel of a signal (SH1) must
    come out as the restatement's, which filters it with EXACTLY the first texel's weights / footprints / blend factors / clamp ratio.
    PrePass, TemporalAccumulation, HistoryFix, A-trous iterations 0 and 1, default build flavour; each fed the oracle's own planes"""
    D = api.Denoiser
    den = int(D.RELAX_DIFFUSE_SPECULAR_SH)
    scene = pkg.synth.Scene(W, H, dolly=0.03)
    hz = pkg.harness.Harness(oracle, [D.RELAX_DIFFUSE_SPECULAR_SH], W, H)
    st = api.RelaxSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1)
    rb = api.ReblurSettings()
    s = dict(maxAccumulatedFrameNum=st.diffuseMaxAccumulatedFrameNum, maxFastAccumulatedFrameNum=st.diffuseMaxFastAccumulatedFrameNum,
             minMaterialForDiffuse=st.minMaterialForDiffuse, minMaterialForSpecular=st.minMaterialForSpecular, roughnessFraction=st.roughnessFraction,
             lobeAngleFraction=st.lobeAngleFraction, planeDistanceSensitivity=rb.planeDistanceSensitivity, historyFixFrameNum=st.historyFixFrameNum,
             historyFixBasePixelStride=st.historyFixBasePixelStride, fastHistoryClampingSigmaScale=st.fastHistoryClampingSigmaScale,
             responsiveRoughnessThreshold=rb.responsiveAccumulationSettings.roughnessThreshold,
             responsiveMinAccum=float(rb.responsiveAccumulationSettings.minAccumulatedFrameNum),
             minHitDistanceWeight=st.minHitDistanceWeight, diffusePrepassBlurRadius=st.diffusePrepassBlurRadius,
             specularPrepassBlurRadius=st.specularPrepassBlurRadius, hitDistanceParameters=(1.0, 0.0, 1.0, 0.0))
    sa = {k: getattr(st, k) for k in ("depthThreshold", "spatialVarianceEstimationHistoryThreshold", "specularVarianceBoost", "diffusePhiLuminance",
                                      "specularPhiLuminance", "diffuseMinLuminanceWeight", "specularMinLuminanceWeight", "lobeAngleFraction",
                                      "specularLobeAngleSlack", "luminanceEdgeStoppingRelaxation", "normalEdgeStoppingRelaxation",
                                      "roughnessEdgeStoppingRelaxation", "roughnessFraction", "enableRoughnessEdgeStopping", "minMaterialForDiffuse",
                                      "minMaterialForSpecular")}
    for g in range(f):
        fr = with_sh1(frame(g))
        hz.frame(scene.common_settings(api, fr, g, reset=(g == 0)), hz.upload(fr), {D.RELAX_DIFFUSE_SPECULAR_SH: st})
    fr, prev = with_sh1(frame(f)), frame(f - 1)
    cs = scene.common_settings(api, fr, f)
    hz.nrd.new_frame()
    hz.nrd.set_common_settings(cs)
    hz.bind(hz.upload(fr))
    hz.nrd.set_denoiser_settings(den, st)
    names = [d["name"].split("::")[1] for d in hz.nrd.dispatches([den])]
    assert names[:6] == ["ClassifyTiles", "PrePass", "TemporalAccumulation", "HistoryFix", "Atrous0", "Atrous1"]
    cur, old = ("_A", "_B") if f % 2 == 0 else ("_B", "_A")
    # 32-byte texels: [signal][SH0 | SH1][4 x fp16]
    rad = lambda name: hz.pool(name).copy().view(np.float16).reshape(H, W, 2, 2, 4)
    lum = lambda name: hz.pool(name).copy().view(np.float16).reshape(H, W, 2)
    u16 = lambda name: hz.pool(name).copy().view(np.uint16).reshape(H, W)
    c = tmp.Consts(fr, W, H, cs.denoisingRange, cs.disocclusionThreshold)
    gcur = ind.decode_guide(fr["viewz"], fr["normal_roughness"])
    gprev = ind.decode_guide(prev["viewz"], prev["normal_roughness"])

    # ---- PrePass
    hz.nrd.denoise_range([den], 0, 2)
    tmp1 = rad("RELAX::Tmp1")
    track = hz.pool("RELAX::SpecHitDistForTracking").copy().view(np.float16).reshape(H, W)
    w0, _, w1 = ind.prepass(fr["viewz"], fr["normal_roughness"], fr["diff"], fr["spec"], fr["view_to_clip"], fr["world_to_view"], cs.frameIndex, cs.denoisingRange,
                            s, exp_hit_weight=True, angle_normal_weight=True, relax_in=True, sh1=(fr["diff_sh1"], fr["spec_sh1"]))
    agree_texel("RELAX-SH PrePass SH0", tmp1[:, :, :, 0], w0, 0.9995)
    agree_texel("RELAX-SH PrePass SH1", tmp1[:, :, :, 1], w1, 0.999, scale_from=w0)

    # ---- TemporalAccumulation
    hist, fast_prev, mom_prev, speeds_prev = rad("RELAX::History"), lum("RELAX::FastHistory" + old), lum("RELAX::Moments" + old), u16("RELAX::HistoryLength" + old)
    hz.nrd.denoise_range([den], 2, 1)
    tmp2, fast, mom, speeds_tmp = rad("RELAX::Tmp2"), lum("RELAX::FastHistory" + cur), lum("RELAX::Moments" + cur), u16("RELAX::HistoryLength_Tmp")
    data2 = hz.pool("RELAX::Data2").copy().view(np.uint32).reshape(H, W)
    rx = dict(moments_prev=mom_prev, max_a_spec=st.specularMaxAccumulatedFrameNum, max_fast_spec=st.specularMaxFastAccumulatedFrameNum, rec709=True,
              sh1_in=np.ascontiguousarray(tmp1[:, :, :, 1]), sh1_hist=np.ascontiguousarray(hist[:, :, :, 1]))
    w_tmp2, w_fast, w_speeds, w_data2, info = tmp.temporal_accumulation(c, s, gcur, gprev, fr["mv"], np.ascontiguousarray(tmp1[:, :, :, 0]),
                                                                         np.ascontiguousarray(hist[:, :, :, 0]), fast_prev, speeds_prev, track, fr["confidence"], True, relax=rx)
    agree("RELAX-SH TA SH0", tmp2[:, :, :, 0], w_tmp2, 1.0, max_ulp=1)
    agree_texel("RELAX-SH TA SH1", tmp2[:, :, :, 1], info["sh1"], 0.999, scale_from=w_tmp2)  # (direction x luminance: components of both signs - cancelling blends, as in the PrePass)
    agree("RELAX-SH TA fast history", fast, w_fast, 1.0, max_ulp=1)

    # ---- HistoryFix
    hz.nrd.denoise_range([den], 3, 1)
    new_hist, speeds_cur = rad("RELAX::History"), u16("RELAX::HistoryLength" + cur)
    al = st.antilagSettings
    rh = dict(moments=mom, normal_power=st.historyFixEdgeStoppingNormalPower, accel=al.accelerationAmount, spatial=al.spatialSigmaScale,
              temporal=al.temporalSigmaScale, reset=al.resetAmount, max_fast_spec=st.specularMaxFastAccumulatedFrameNum, rec709=True,
              sh1=np.ascontiguousarray(tmp2[:, :, :, 1]))
    w_sig, w_speeds_cur, _, w_sig1 = tmp.history_fix(c, s, gcur, np.ascontiguousarray(tmp2[:, :, :, 0]), speeds_tmp, fast, fr["viewz"], fr["normal_roughness"],
                                                     upstream=True, relax=rh)
    agree("RELAX-SH HistoryFix SH0", new_hist[:, :, :, 0], w_sig, 1.0, max_ulp=1)
    agree_texel("RELAX-SH HistoryFix SH1", new_hist[:, :, :, 1], w_sig1, 0.999, scale_from=w_sig)

    # ---- A-trous iterations 0 and 1
    moments = lum("RELAX::Moments" + cur)
    hz.nrd.denoise_range([den], 4, 1)
    a0 = rad("RELAX::Atrous_A")
    w_a0, w_a0_1 = tmp.atrous_iteration(c, sa, gcur, np.ascontiguousarray(new_hist[:, :, :, 0]), 0, speeds_cur, moments, data2, upstream=True,
                                        sh1=np.ascontiguousarray(new_hist[:, :, :, 1]))
    agree("RELAX-SH A-trous 0 SH0", a0[:, :, :, 0], w_a0, 1.0, max_ulp=1)
    agree_texel("RELAX-SH A-trous 0 SH1", a0[:, :, :, 1], w_a0_1, 0.999, scale_from=w_a0[..., :3])
    hz.nrd.denoise_range([den], 5, 1)
    a1 = rad("RELAX::Atrous_B")
    w_a1, w_a1_1 = tmp.atrous_iteration(c, sa, gcur, np.ascontiguousarray(a0[:, :, :, 0]), 1, data2=data2, upstream=True, sh1=np.ascontiguousarray(a0[:, :, :, 1]))
    agree("RELAX-SH A-trous 1 SH0", a1[:, :, :, 0], w_a1, 1.0, max_ulp=1)
    agree_texel("RELAX-SH A-trous 1 SH1", a1[:, :, :, 1], w_a1_1, 0.999, scale_from=w_a1[..., :3])


@pytest.mark.parametrize("f", [1, 2, 3])
def test_sigma_passes_independent(pkg, api, oracle, f):
    """SIGMA_SHADOW_TRANSLUCENCY: Blur, PostBlur (penumbra-sized tangent-plane blur of the visibility) and TemporalStabilization
    (reprojection with occlusion test, per-channel 5x5 moment clamp, sqrt-encoded RGBA8 history)"""
    D = api.Denoiser
    den = int(D.SIGMA_SHADOW_TRANSLUCENCY)
    scene = pkg.synth.Scene(W, H, dolly=0.03)
    hz = pkg.harness.Harness(oracle, [D.SIGMA_SHADOW_TRANSLUCENCY], W, H)
    st = api.SigmaSettings(lightDirection=list(scene.sun))
    s = dict(planeDistanceSensitivity=st.planeDistanceSensitivity, maxStabilizedFrameNum=st.maxStabilizedFrameNum)
    for g in range(f):
        fr = frame(g)
        hz.frame(scene.common_settings(api, fr, g, reset=(g == 0)), hz.upload(fr), {D.SIGMA_SHADOW_TRANSLUCENCY: st})
    fr, prev = frame(f), frame(f - 1)
    cs = scene.common_settings(api, fr, f)
    hz.nrd.new_frame()
    hz.nrd.set_common_settings(cs)
    hz.bind(hz.upload(fr))
    hz.nrd.set_denoiser_settings(den, st)
    names = [d["name"].split("::")[1] for d in hz.nrd.dispatches([den])]
    assert names == ["ClassifyTiles", "SmoothTiles", "Blur", "PostBlur", "TemporalStabilization"]
    cur, old = ("_A", "_B") if f % 2 == 0 else ("_B", "_A")
    c = tmp.Consts(fr, W, H, cs.denoisingRange, cs.disocclusionThreshold)
    gcur = ind.decode_guide(fr["viewz"], fr["normal_roughness"])
    gprev = ind.decode_guide(prev["viewz"], prev["normal_roughness"])
    z, n = gcur[0], gcur[1]
    # ---- ClassifyTiles and SmoothTiles (round 6): the per-tile flags / radii every later pass of the frame is steered by - exact
    hz.nrd.denoise_range([den], 0, 1)
    raw_tiles = hz.pool("SIGMA::Tiles").copy().view(np.uint16).reshape((H + 15) // 16, (W + 15) // 16)
    w_tiles = tmp.sigma_classify_tiles(c, z, fr["penumbra"])
    assert np.array_equal(raw_tiles & 3, w_tiles & 3), "ClassifyTiles: penumbra / lit flags"
    dr = np.abs((raw_tiles >> 8).astype(np.int32) - (w_tiles >> 8).astype(np.int32))
    assert dr.max() <= 1 and float((dr == 0).mean()) >= 0.9, "ClassifyTiles: tile radius (a float32 quotient next to an integer may round up the other way)"
    hz.nrd.denoise_range([den], 1, 1)
    tiles = hz.pool("SIGMA::SmoothTiles").copy().view(np.uint16).reshape((H + 15) // 16, (W + 15) // 16)
    assert np.array_equal(tiles, tmp.sigma_smooth_tiles(raw_tiles)), "SmoothTiles (fed the oracle's Tiles plane)"
    assert (tiles & 1).any() and (raw_tiles & 3 == 3).any(), "the scene must have penumbra tiles for this test to mean anything"
    vis = tmp.sigma_input_visibility(fr["penumbra"].astype(np.float64), fr["translucency"])
    w_sh1, w_pen1 = tmp.sigma_blur(c, s, z, n, tiles, fr["penumbra"], vis, cs.frameIndex, 0)
    hz.nrd.denoise_range([den], 2, 1)
    sh1 = hz.pool("SIGMA::Shadow1").copy().view(np.float16).reshape(H, W, 4)
    pen1 = hz.pool("SIGMA::Penumbra1").copy().view(np.float16).reshape(H, W)
    agree("SIGMA Blur shadow", sh1, w_sh1, 1.0, max_ulp=1)
    agree("SIGMA Blur penumbra", pen1, w_pen1, 1.0, max_ulp=1)
    w_sh2, _ = tmp.sigma_blur(c, s, z, n, tiles, pen1, sh1, cs.frameIndex, 1)
    hz.nrd.denoise_range([den], 3, 1)
    sh2 = hz.pool("SIGMA::Shadow2").copy().view(np.float16).reshape(H, W, 4)
    agree("SIGMA PostBlur shadow", sh2, w_sh2, 1.0, max_ulp=1)
    hist_prev = hz.pool("SIGMA::History" + old).copy().view(np.uint32).reshape(H, W)
    hz.nrd.denoise_range([den], 4, 1)
    hist = hz.pool("SIGMA::History" + cur).copy().view(np.uint8).reshape(H, W, 4).astype(np.int32)
    want = tmp.sigma_temporal_stabilization(c, s, gcur, gprev, fr["mv"], sh2, tiles, hist_prev, True)
    want = np.stack([(want >> (8 * i)) & 255 for i in range(4)], -1).astype(np.int32)
    assert float((np.abs(hist - want) <= 1).mean()) > 0.99 and float((hist == want).mean()) > 0.97


@pytest.mark.parametrize("recon,radius", [(None, 0), ("AREA_3X3", 1), ("AREA_5X5", 2)])
def test_prepare_inputs_independent(pkg, api, oracle, recon, radius):
    """PrepareInputs at the sample's default operating point (checkerboard WHITE: both signals on complementary colours of half-width
    inputs) with and without the hit-distance reconstruction, 40 % of the hit distances punched out: the dense planes the oracle hands to
    the PrePass against tests/indep/prepare_numpy.py, frames 0 and 1 (the checkerboard phase flips with the frame index)"""
    import prepare_numpy as prep

    D = api.Denoiser
    den = int(D.REBLUR_DIFFUSE_SPECULAR)
    scene = pkg.synth.Scene(W, H, dolly=0.03)
    hz = pkg.harness.Harness(oracle, [D.REBLUR_DIFFUSE_SPECULAR], W, H)
    st = api.ReblurSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1, checkerboardMode=int(api.CheckerboardMode.WHITE),
                            hitDistanceReconstructionMode=int(getattr(api.HitDistanceReconstructionMode, recon or "OFF")))
    s = dict(minMaterialForDiffuse=0, minMaterialForSpecular=1, lobeAngleFraction=st.lobeAngleFraction, roughnessFraction=st.roughnessFraction)
    rng = np.random.default_rng(5)
    for f in range(2):
        fr = frame(f)
        for key in ("diff", "spec"):
            a = np.array(fr[key])
            a[rng.random(a.shape[:2]) < 0.4, 3] = 0
            fr[key] = a
        fr.update(pkg.harness.to_checkerboard(fr, f, white=True))
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        hz.nrd.new_frame()
        hz.nrd.set_common_settings(cs)
        hz.bind(hz.upload(fr))
        hz.nrd.set_denoiser_settings(den, st)
        names = [d["name"].split("::")[1] for d in hz.nrd.dispatches([den])]
        assert names[:2] == ["ClassifyTiles", "PrepareInputs"]
        hz.nrd.denoise_range([den], 0, 2)
        geo = ind.plane_terms(fr["viewz"], fr["normal_roughness"], fr["view_to_clip"], fr["world_to_view"], st.planeDistanceSensitivity)
        for key, plane, phase, is_spec in (("diff", "REBLUR::Prepared_Diff", 1, False), ("spec", "REBLUR::Prepared_Spec", 0, True)):
            got = hz.pool(plane).copy().view(np.float16).reshape(H, W, 4)
            half = np.asarray(fr[key]).view(np.float16).reshape(H, -1, 4)
            want = prep.prepare_inputs(fr["viewz"], fr["normal_roughness"], half, phase, cs.frameIndex, cs.denoisingRange, is_spec, radius, geo, s)
            agree("PrepareInputs %s frame %d" % (key, f), got, want, 0.9998, max_ulp=2)
        hz.nrd.denoise_range([den], 2, len(names) - 2)  # finish the frame: the next one starts from a consistent instance
