"""Settings the sample's UI can flip (Source/NRDSample.cpp:1515-1663) beyond the defaults: anti-firefly, dynamic resolution
(rectSize < resourceSize changing between frames, :3847-3854), history-confidence inputs - known answers on the oracle, bit-exact
emulated kernels on CPU, bit-exact HIP on the GPU."""
import numpy as np
import pytest

import util


def test_anti_firefly_suppresses_outlier(pkg, api, oracle):
    D = api.Denoiser
    w, h = 48, 32
    for den, mk in ((D.REBLUR_DIFFUSE, api.ReblurSettings), (D.RELAX_DIFFUSE, api.RelaxSettings)):
        res = {}
        for on in (False, True):
            hz = pkg.harness.Harness(oracle, [den], w, h)
            st = {den: mk(enableAntiFirefly=on, diffusePrepassBlurRadius=0.0)}
            fr = util.flat_frame(pkg, w, h)
            if den == D.RELAX_DIFFUSE:
                fr["diff"][..., :3] = np.float16(0.5)
            fr["diff"][16, 24, 0] = 60.0  # one firefly (luma channel of REBLUR's YCoCg / red of RELAX's RGB)
            hz.frame(util.static_common(api, w, h, reset=True), hz.upload(fr), st)
            res[on] = hz.output("out_diff")[12:21, 20:29, 0].astype(np.float32).max()
        base = 0.5 if den == D.RELAX_DIFFUSE else float(util.flat_frame(pkg, w, h)["diff"][0, 0, 0])
        assert abs(res[True] - base) < 0.5 * abs(res[False] - base), (den, res)  # the firefly's footprint is at least halved


def run_pair(pkg, api, a, b, dens, frames, st, common_hook=None, frame_hook=None):
    w, h = 60, 44
    scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].name.startswith("RELAX") else "REBLUR")
    ha = util.run_frames(api, pkg.harness, a, scene, dens, frames, settings=st(scene), common_hook=common_hook, frame_hook=frame_hook)
    hb = util.run_frames(api, pkg.harness, b, scene, dens, frames, settings=st(scene), common_hook=common_hook, frame_hook=frame_hook)
    return util.compare_all(ha, hb, exact=True)


def drs_hook(f, cs):
    """dynamic resolution: the rect shrinks and grows inside a fixed resource (Source/NRDSample.cpp:3847-3854)"""
    sizes = [(60, 44), (48, 36), (54, 40), (60, 44)]
    w, h = sizes[f % 4]
    pw, ph = sizes[(f - 1) % 4] if f > 0 else (w, h)
    cs.rectSize[0], cs.rectSize[1] = w, h
    cs.rectSizePrev[0], cs.rectSizePrev[1] = pw, ph
    cs.motionVectorScale[0], cs.motionVectorScale[1] = 1.0 / w, 1.0 / h


def conf_hook(f, cs):
    cs.isHistoryConfidenceAvailable = True


def conf_frames(f, fr):
    c = np.array(fr["confidence"], copy=True)
    c[..., 0] = np.float16(0.25) + np.float16(0.5) * (np.arange(c.shape[1], dtype=np.float32) / c.shape[1]).astype(np.float16)[None, :]
    fr["confidence"] = c


CASES = [
    ("antifirefly_reblur", ["REBLUR_DIFFUSE_SPECULAR"], dict(enableAntiFirefly=True), None, None),
    ("antifirefly_relax_sh", ["RELAX_DIFFUSE_SPECULAR_SH"], dict(enableAntiFirefly=True), None, None),
    ("drs_reblur_sigma", ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW_TRANSLUCENCY"], {}, drs_hook, None),
    ("drs_relax", ["RELAX_DIFFUSE_SPECULAR"], {}, drs_hook, None),
    ("confidence_reblur", ["REBLUR_DIFFUSE_SPECULAR"], {}, conf_hook, conf_frames),
    ("confidence_relax", ["RELAX_DIFFUSE_SPECULAR"], {}, conf_hook, conf_frames),
    ("validation_reblur", ["REBLUR_DIFFUSE_SPECULAR"], {}, lambda f, cs: setattr(cs, "enableValidation", True), None),
    ("jitter_reblur_sigma", ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW"], {}, lambda f, cs: jitter_hook(f, cs), None),
    ("strand_reblur", ["REBLUR_DIFFUSE_SPECULAR"], {}, lambda f, cs: strand_hook(f, cs), None),
    ("disocclusion_mix", ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW"], {}, lambda f, cs: mix_hook(f, cs), lambda f, fr: mix_frames(f, fr)),
    ("disocclusion_mix_relax", ["RELAX_DIFFUSE"], {}, lambda f, cs: mix_hook(f, cs), lambda f, fr: mix_frames(f, fr)),
    ("strand_relax_sh", ["RELAX_DIFFUSE_SPECULAR_SH"], {}, lambda f, cs: strand_hook(f, cs), None),
    # RELAX tuning fields of the sample's UI (Source/NRDSample.cpp:1600-1606 antilag, :1626 history-fix normal power, :1650 relaxation)
    ("relax_tuning", ["RELAX_DIFFUSE_SPECULAR"], {"luminanceEdgeStoppingRelaxation": 1.0, "normalEdgeStoppingRelaxation": 0.8,
                                                  "roughnessEdgeStoppingRelaxation": 0.4, "historyFixEdgeStoppingNormalPower": 3.0,
                                                  "antilagSettings.accelerationAmount": 0.8, "antilagSettings.spatialSigmaScale": 0.5,
                                                  "antilagSettings.temporalSigmaScale": 0.1, "antilagSettings.resetAmount": 1.0}, None, None),
    ("prepass_track_only", ["REBLUR_DIFFUSE_SPECULAR_SH"], {"usePrepassOnlyForSpecularMotionEstimation": True}, None, None),
    ("history_length_out", ["REBLUR_DIFFUSE_SPECULAR_OCCLUSION"], {"returnHistoryLengthInsteadOfOcclusion": True}, None, None),
    ("relax_confidence_driven", ["RELAX_DIFFUSE_SPECULAR"], {"confidenceDrivenRelaxationMultiplier": 2.0, "confidenceDrivenLuminanceEdgeStoppingRelaxation": 0.8,
                                                            "confidenceDrivenNormalEdgeStoppingRelaxation": 0.5, "specularLobeAngleSlack": 1.0}, conf_hook, conf_frames),
    ("camera_attached_reblur", ["REBLUR_DIFFUSE_SPECULAR"], {}, lambda f, cs: attach_hook(f, cs), None),
    ("camera_attached_relax", ["RELAX_SPECULAR"], {}, lambda f, cs: attach_hook(f, cs), None),
    # non-default hit-distance normalisation (the sample's "HitT scale" slider feeds A, Source/NRDSample.cpp:3675; C / D shape the
    # roughness term whose diffuse value the host precomputes: ReblurParams::hitFactorDiff)
    ("hit_distance_parameters", ["REBLUR_DIFFUSE_SPECULAR"], {"hitDistanceParameters.A": 1.7, "hitDistanceParameters.B": 0.25,
                                                              "hitDistanceParameters.C": 12.0, "hitDistanceParameters.D": -13.5}, None, None),
    ("relax_tuning_sh", ["RELAX_SPECULAR_SH"], {"luminanceEdgeStoppingRelaxation": 0.0, "normalEdgeStoppingRelaxation": 1.0,
                                                "antilagSettings.resetAmount": 0.0, "antilagSettings.accelerationAmount": 0.0}, None, None),
]


def strand_hook(f, cs):
    """the sample's hair settings (Source/NRDSample.cpp:3871-3872); material 1 plays the strand here (the scene's spheres carry it)"""
    cs.strandMaterialID = 1.0
    cs.strandThickness = 0.002


def attach_hook(f, cs):
    """CommonSettings::cameraAttachedReflectionMaterialID (Source/NRDSample.cpp:3869-3876); material 1 (the scene's spheres) plays
    the surface that shows reflections of camera-attached objects"""
    cs.cameraAttachedReflectionMaterialID = 1.0


def mix_hook(f, cs):
    cs.isDisocclusionThresholdMixAvailable = True
    cs.disocclusionThresholdAlternate = 0.0005


def mix_frames(f, fr):
    h, w = np.asarray(fr["viewz"]).shape[:2]
    fr["disocclusion_mix"] = np.tile((np.arange(w) * 255 // max(w - 1, 1)).astype(np.uint8)[None, :], (h, 1))  # 0 left .. 1 right


def jitter_hook(f, cs):
    halton = [(0.0, -0.1667), (-0.25, 0.1667), (0.25, -0.3889), (-0.375, -0.0556), (0.125, 0.2778)]
    cs.cameraJitter[0], cs.cameraJitter[1] = halton[f % 5]
    cs.cameraJitterPrev[0], cs.cameraJitterPrev[1] = halton[(f - 1) % 5] if f > 0 else halton[0]



def settings_factory(api, dens, kw):
    def make(scene):
        st = util.default_settings(api, scene, dens, minMaterialForDiffuse=0, minMaterialForSpecular=1)
        for d in dens:
            if d.name.startswith("REBLUR") or d.name.startswith("RELAX"):
                for k, v in kw.items():
                    obj = st[d]
                    *path, leaf = k.split(".")
                    for part in path:
                        obj = getattr(obj, part)
                    assert hasattr(obj, leaf), k
                    setattr(obj, leaf, v)
        return st
    return make


@pytest.mark.parametrize("name,dens,kw,chook,fhook", CASES, ids=[c[0] for c in CASES])
def test_variants_emulated_bit_exact(pkg, api, oracle, emulated, name, dens, kw, chook, fhook):
    dd = [api.Denoiser[x] for x in dens]
    assert run_pair(pkg, api, oracle, emulated, dd, 3, settings_factory(api, dd, kw), chook, fhook) == []


@pytest.mark.gpu
@pytest.mark.parametrize("name,dens,kw,chook,fhook", CASES, ids=[c[0] for c in CASES])
def test_variants_hip_bit_exact(pkg, api, oracle, hip, name, dens, kw, chook, fhook):
    dd = [api.Denoiser[x] for x in dens]
    assert run_pair(pkg, api, oracle, hip, dd, 4, settings_factory(api, dd, kw), chook, fhook) == []


def validation_hook(f, cs):
    cs.enableValidation = True


def test_validation_overlay(pkg, api, oracle, emulated):
    """CommonSettings::enableValidation (Source/NRDSample.cpp:3867) adds a Validation dispatch writing OUT_VALIDATION (:452)"""
    D = api.Denoiser
    for dens in ([D.REBLUR_DIFFUSE_SPECULAR], [D.RELAX_DIFFUSE_SPECULAR]):
        w, h = 60, 44
        scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].name.startswith("RELAX") else "REBLUR")
        st = util.default_settings(api, scene, dens, minMaterialForDiffuse=0, minMaterialForSpecular=1)
        outs = []
        for b in (oracle, emulated):
            hz = util.run_frames(api, pkg.harness, b, scene, dens, 4, settings=st, common_hook=validation_hook)
            names = [x["name"] for x in hz.nrd.dispatches([int(dens[0])])]
            assert names[-1].endswith("::Validation")
            outs.append(hz.fetch(hz.outputs["out_validation"]).reshape(h, w, 4).copy())
        assert np.array_equal(outs[0], outs[1])
        v = outs[0]
        sky = scene.frame(3)["viewz"] > 1e4
        assert np.all(v[sky] == 0) and v[~sky][:, 0].max() >= int(3 / 63 * 255)  # accumulated frames show up, sky stays black
        # off by default: no such dispatch
        hz = util.run_frames(api, pkg.harness, oracle, scene, dens, 1, settings=st)
        assert not any(x["name"].endswith("Validation") for x in hz.nrd.dispatches([int(dens[0])]))


def test_camera_jitter_is_a_pure_pixel_grid_shift(pkg, api, oracle):
    """cameraJitter (Source/NRDSample.cpp:3843-3846): a static camera with the SAME jitter in both frames leaves a flat scene a
    fixed point; a jitter difference between the frames shifts the reprojection by exactly that many pixels."""
    D = api.Denoiser
    w, h = 48, 32
    dens = [D.REBLUR_DIFFUSE_SPECULAR]
    hz = pkg.harness.Harness(oracle, dens, w, h)
    for f in range(3):
        fr = util.flat_frame(pkg, w, h)
        cs = util.static_common(api, w, h, frame_index=f, reset=(f == 0))
        cs.cameraJitter[0], cs.cameraJitter[1] = 0.3, -0.2
        cs.cameraJitterPrev[0], cs.cameraJitterPrev[1] = 0.3, -0.2
        hz.frame(cs, hz.upload(fr), {dens[0]: api.ReblurSettings()})
        assert util.max_ulp_f16(hz.output("out_diff"), fr["diff"]) <= 1
    # a world-space-static surface seen through a jitter that moved by (+1, 0) pixel reprojects one pixel to the side:
    # with 2-D motion vectors of zero the reprojected uv is the same pixel, so nothing breaks either (smoke check of the prev path)
    fr = util.flat_frame(pkg, w, h)
    cs = util.static_common(api, w, h, frame_index=3)
    cs.cameraJitter[0], cs.cameraJitterPrev[0] = 0.5, -0.5
    hz.frame(cs, hz.upload(fr), {dens[0]: api.ReblurSettings()})
    assert util.max_ulp_f16(hz.output("out_diff"), fr["diff"]) <= 1


RELAX_FIELDS = [
    {"luminanceEdgeStoppingRelaxation": 0.0}, {"normalEdgeStoppingRelaxation": 1.0}, {"roughnessEdgeStoppingRelaxation": 0.0},
    {"historyFixEdgeStoppingNormalPower": 1.0}, {"antilagSettings.accelerationAmount": 1.0}, {"specularLobeAngleSlack": 3.0},
    {"antilagSettings.resetAmount": 1.0, "antilagSettings.spatialSigmaScale": 0.25, "antilagSettings.temporalSigmaScale": 0.0},
]


@pytest.mark.parametrize("kw", RELAX_FIELDS, ids=[next(iter(k)).split(".")[-1] for k in RELAX_FIELDS])
def test_relax_tuning_fields_are_honoured(pkg, api, oracle, kw):
    """every RELAX tuning field the sample's UI exposes changes the result under camera motion (none is silently ignored)"""
    den = api.Denoiser.RELAX_DIFFUSE_SPECULAR
    w, h = 60, 44
    scene = pkg.synth.Scene(w, h, dolly=0.06, denoiser="RELAX")
    fast = {"diffuseMaxFastAccumulatedFrameNum": 1, "specularMaxFastAccumulatedFrameNum": 1}  # the clamp acts from frame 2 on
    base = util.run_frames(api, pkg.harness, oracle, scene, [den], 4, settings=settings_factory(api, [den], fast)(scene))
    var = util.run_frames(api, pkg.harness, oracle, scene, [den], 4, settings=settings_factory(api, [den], {**fast, **kw})(scene))
    diffs = util.compare_all(base, var, exact=True)
    assert diffs != [], kw


def test_relax_antilag_reset_speeds_up_a_lighting_change(pkg, api, oracle):
    """RelaxAntilagSettings: after an abrupt brightness change a full reset (resetAmount 1, tight sigma scales) follows the new
    signal faster than antilag off (resetAmount 0, accelerationAmount 0)"""
    den = api.Denoiser.RELAX_DIFFUSE
    w, h = 48, 32
    out = {}
    for name, al in (("off", dict(resetAmount=0.0, accelerationAmount=0.0)),
                     ("on", dict(resetAmount=1.0, accelerationAmount=1.0, spatialSigmaScale=0.5, temporalSigmaScale=0.0))):
        hz = pkg.harness.Harness(oracle, [den], w, h)
        st = api.RelaxSettings(diffusePrepassBlurRadius=0.0, diffuseMaxAccumulatedFrameNum=30, diffuseMaxFastAccumulatedFrameNum=2)
        for k, v in al.items():
            setattr(st.antilagSettings, k, v)
        for f in range(12):
            fr = util.flat_frame(pkg, w, h)
            fr["diff"][..., :3] = np.float16(0.2 if f < 10 else 1.0)
            hz.frame(util.static_common(api, w, h, reset=(f == 0)), hz.upload(fr), {den: st})
        out[name] = float(hz.output("out_diff")[8:24, 8:40, 0].astype(np.float32).mean())
    assert out["on"] > out["off"] + 0.05, out
    assert out["on"] <= 1.0 + 1e-3


def test_reblur_flags_semantics(pkg, api, oracle):
    """usePrepassOnlyForSpecularMotionEstimation leaves the diffuse path alone and changes the specular one;
    returnHistoryLengthInsteadOfOcclusion reports accumulated frames / maxAccumulatedFrameNum in OUT_*_HITDIST"""
    D = api.Denoiser
    w, h = 60, 44
    scene = pkg.synth.Scene(w, h, dolly=0.0)
    base = util.run_frames(api, pkg.harness, oracle, scene, [D.REBLUR_DIFFUSE_SPECULAR], 1, settings=settings_factory(api, [D.REBLUR_DIFFUSE_SPECULAR], {})(scene))
    var = util.run_frames(api, pkg.harness, oracle, scene, [D.REBLUR_DIFFUSE_SPECULAR], 1,
                          settings=settings_factory(api, [D.REBLUR_DIFFUSE_SPECULAR], {"usePrepassOnlyForSpecularMotionEstimation": True})(scene))
    assert np.array_equal(base.output("out_diff").view(np.uint16), var.output("out_diff").view(np.uint16))
    assert not np.array_equal(base.output("out_spec").view(np.uint16), var.output("out_spec").view(np.uint16))

    den = D.REBLUR_DIFFUSE_SPECULAR_OCCLUSION
    keep = []
    st = settings_factory(api, [den], {"returnHistoryLengthInsteadOfOcclusion": True, "maxAccumulatedFrameNum": 20})(scene)
    util.run_frames(api, pkg.harness, oracle, scene, [den], 6, settings=st, keep=keep)
    z = np.asarray(scene.frame(0)["viewz"], dtype=np.float32)
    m = z < 1e4
    med = [float(np.median(np.asarray(k["out_diff_hitdist"]).view(np.uint16).reshape(h, w).astype(np.float32)[m] / 65535.0)) for k in keep]
    assert med[0] == 0.0 and all(b > a for a, b in zip(med, med[1:])), med  # one more frame per frame, 0 on the first
    assert abs(med[5] - 5.0 / 20.0) < 0.02, med


def test_relax_confidence_driven_relaxation(pkg, api, oracle):
    """confidenceDriven* only act with confidence inputs below 1; then they change the A-trous result"""
    den = api.Denoiser.RELAX_DIFFUSE_SPECULAR
    scene = pkg.synth.Scene(60, 44, dolly=0.06, denoiser="RELAX")
    kw = {"confidenceDrivenRelaxationMultiplier": 2.0, "confidenceDrivenLuminanceEdgeStoppingRelaxation": 1.0, "confidenceDrivenNormalEdgeStoppingRelaxation": 1.0}
    run = lambda k, ch, fh: util.run_frames(api, pkg.harness, oracle, scene, [den], 3, settings=settings_factory(api, [den], k)(scene), common_hook=ch, frame_hook=fh)
    assert util.compare_all(run({}, None, None), run(kw, None, None), exact=True) == []  # no confidence inputs: no effect
    assert util.compare_all(run({}, conf_hook, None), run(kw, conf_hook, None), exact=True) == []  # confidence == 1 everywhere: no effect
    assert util.compare_all(run({}, conf_hook, conf_frames), run(kw, conf_hook, conf_frames), exact=True) != []


def test_strand_material_relaxes_only_its_pixels(pkg, api, oracle):
    """strandMaterialID / strandThickness change the result on pixels of that material (and their filter neighbourhood) only"""
    den = api.Denoiser.REBLUR_DIFFUSE_SPECULAR
    scene = pkg.synth.Scene(96, 64, dolly=0.0)
    st = settings_factory(api, [den], {})(scene)
    base = util.run_frames(api, pkg.harness, oracle, scene, [den], 2, settings=st)
    var = util.run_frames(api, pkg.harness, oracle, scene, [den], 2, settings=st, common_hook=strand_hook)
    a, b = base.output("out_diff").view(np.uint16), var.output("out_diff").view(np.uint16)
    changed = (a != b).any(-1)
    nr = np.asarray(scene.frame(0)["normal_roughness"]).view(np.uint32).reshape(64, 96)
    mat1 = (nr >> 30) == 1
    assert changed.any() and changed[mat1].mean() > 0.5
    from scipy.ndimage import binary_dilation
    near = binary_dilation(mat1, iterations=60)  # farther than the widest filter reach from any strand pixel: bit-identical
    assert not changed[~near].any()


def test_camera_attached_reflection_material(pkg, api, oracle):
    """cameraAttachedReflectionMaterialID re-aims the virtual-motion reprojection of the specular signal on pixels of that
    material only: with a moving camera those pixels (and what the spatial passes spread from them) change, the diffuse signal
    does not; a material nobody carries and the default (999) leave every plane bit-identical; a static camera makes it a no-op
    up to rounding"""
    den = api.Denoiser.REBLUR_DIFFUSE_SPECULAR
    w, h = 96, 64
    scene = pkg.synth.Scene(w, h, dolly=0.05)
    st = settings_factory(api, [den], {})(scene)
    nr = np.asarray(scene.frame(0)["normal_roughness"]).view(np.uint32).reshape(h, w)
    mats = set(np.unique(nr >> 30).tolist())
    assert 1 in mats and 3 not in mats

    def hook_of(v):
        return lambda f, cs: setattr(cs, "cameraAttachedReflectionMaterialID", v)

    base = util.run_frames(api, pkg.harness, oracle, scene, [den], 4, settings=st)
    var = util.run_frames(api, pkg.harness, oracle, scene, [den], 4, settings=st, common_hook=hook_of(1.0))
    assert util.compare_all(base, util.run_frames(api, pkg.harness, oracle, scene, [den], 4, settings=st, common_hook=hook_of(3.0)), exact=True) == []
    assert util.compare_all(base, util.run_frames(api, pkg.harness, oracle, scene, [den], 4, settings=st, common_hook=hook_of(999.0)), exact=True) == []
    a, b = base.output("out_spec").view(np.uint16), var.output("out_spec").view(np.uint16)
    changed = (a != b).any(-1)
    mat1 = (np.asarray(scene.frame(3)["normal_roughness"]).view(np.uint32).reshape(h, w) >> 30) == 1
    assert changed[mat1].mean() > 0.25, changed[mat1].mean()
    from scipy.ndimage import binary_dilation
    assert not changed[~binary_dilation(mat1, iterations=60)].any()
    assert (base.output("out_diff").view(np.uint16) == var.output("out_diff").view(np.uint16)).all()
    # static camera: both aims are the same point up to the rounding of the reprojection chain
    still = pkg.synth.Scene(w, h, dolly=0.0)
    st0 = settings_factory(api, [den], {})(still)
    s0 = util.run_frames(api, pkg.harness, oracle, still, [den], 4, settings=st0).output("out_spec").astype(np.float32)
    s1 = util.run_frames(api, pkg.harness, oracle, still, [den], 4, settings=st0, common_hook=hook_of(1.0)).output("out_spec").astype(np.float32)
    moving = np.abs(base.output("out_spec").astype(np.float32) - var.output("out_spec").astype(np.float32)).mean()
    assert np.abs(s0 - s1).mean() < 0.02 * moving, (np.abs(s0 - s1).mean(), moving)


def test_disocclusion_threshold_mix(pkg, api, oracle):
    """IN_DISOCCLUSION_THRESHOLD_MIX blends disocclusionThreshold toward disocclusionThresholdAlternate per pixel: a very strict
    alternate threshold on the right half of the image drops history there (moving camera), the left half keeps it; the slot is
    required once isDisocclusionThresholdMixAvailable is set"""
    den = api.Denoiser.REBLUR_DIFFUSE_SPECULAR
    w, h = 96, 64
    scene = pkg.synth.Scene(w, h, dolly=0.05)
    st = settings_factory(api, [den], {})(scene)

    def hook(f, cs):
        cs.enableValidation = True
        cs.isDisocclusionThresholdMixAvailable = True
        cs.disocclusionThresholdAlternate = 0.0

    def frames(f, fr):
        m = np.zeros((h, w), dtype=np.uint8)
        m[:, w // 2:] = 255
        fr["disocclusion_mix"] = m

    hz = util.run_frames(api, pkg.harness, oracle, scene, [den], 5, settings=st, common_hook=hook, frame_hook=frames)
    val = hz.fetch(hz.outputs["out_validation"]).reshape(h, w, 4)[..., 0].astype(np.float32) / 255.0 * 63.0
    z = np.asarray(scene.frame(4)["viewz"], dtype=np.float32)
    hit = z < 1e4
    left, right = hit.copy(), hit.copy()
    left[:, w // 2 - 4:] = False
    right[:, :w // 2 + 4] = False
    assert float(np.median(val[left])) >= 3.0 and float(np.median(val[right])) <= 1.0, (np.median(val[left]), np.median(val[right]))
    with pytest.raises(api.NrdError):
        util.run_frames(api, pkg.harness, oracle, scene, [den], 1, settings=st, common_hook=hook)  # slot not bound
