"""Known-answer tests derivable from the reference's contract alone; the reference itself pins
nothing numerically ("parity unpinned"), so these invariants are what every backend must satisfy. Parametrised over the CPU
oracle, the host-emulated product kernels, and (marked gpu) the product on MI355X through the C-ABI."""
import numpy as np
import pytest

import util

BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]
BACKENDS_EMU = ["oracle", "emulated", pytest.param("hip", marks=pytest.mark.gpu)]


def get(request, name):
    return request.getfixturevalue(name)


def size_for(name, big, small):
    return small if name == "emulated" else big


# ---- (1) REFERENCE ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS_EMU)
def test_reference_accumulation(request, pkg, api, backend):
    b = get(request, backend)
    w, h = size_for(backend, (96, 64), (32, 16))
    D = api.Denoiser
    hz = pkg.harness.Harness(b, [D.REFERENCE], w, h)
    rng = np.random.default_rng(5)
    base = np.zeros((h, w, 4))
    base[..., 0] = (np.arange(w) + 0.5) / w
    base[..., 1] = ((np.arange(h) + 0.5) / h)[:, None]
    base[..., 2] = 0.5
    base[..., 3] = 1.0
    st = {D.REFERENCE: api.ReferenceSettings(maxAccumulatedFrameNum=1024)}
    errs = []
    mean = np.zeros_like(base)
    for f in range(16):
        reset = f in (0, 8)
        noisy = (base * (1 + 0.5 * rng.uniform(-1, 1, base.shape))).astype(np.float16)
        cs = util.static_common(api, w, h, f, reset=reset)
        planes = hz.upload({"signal": noisy})
        hz.frame(cs, planes, st)
        out = hz.fetch(planes["signal"]).view(np.float16).reshape(h, w, 4).astype(np.float64)  # in place (NRDSample.cpp:484-485)
        if reset:
            # CLEAR_AND_RESTART => output = input
            assert np.array_equal(out.astype(np.float16), noisy)
            mean = noisy.astype(np.float64)
            n = 1
        else:
            n += 1
            mean = mean + (noisy.astype(np.float64) - mean) / n
            assert np.abs(out - mean).max() < 2e-3  # fp32 running mean, fp16 output rounding
        errs.append(np.sqrt(np.mean((out[..., :3] - base[..., :3]) ** 2)))
    # error variance ~ 1/N within each accumulation run
    assert errs[7] < errs[0] * 0.5 and errs[15] < errs[8] * 0.5


@pytest.mark.parametrize("backend", BACKENDS)
def test_reference_constant_and_cap(request, pkg, api, backend):
    b = get(request, backend)
    w, h = 48, 32
    D = api.Denoiser
    hz = pkg.harness.Harness(b, [D.REFERENCE], w, h)
    const = np.full((h, w, 4), 0.75, dtype=np.float16)
    st = {D.REFERENCE: api.ReferenceSettings(maxAccumulatedFrameNum=2)}
    for f in range(5):
        planes = hz.upload({"signal": const})
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), planes, st)
        assert np.array_equal(hz.fetch(planes["signal"]).view(np.float16).reshape(h, w, 4), const)
    # cap: with maxAccumulatedFrameNum = 2 a step input converges geometrically with weight 1/3
    step = np.full((h, w, 4), 1.5, dtype=np.float16)
    planes = hz.upload({"signal": step})
    hz.frame(util.static_common(api, w, h, 5), planes, st)
    got = hz.fetch(planes["signal"]).view(np.float16).reshape(h, w, 4).astype(np.float32)
    assert np.allclose(got, 0.75 + (1.5 - 0.75) / 3.0, atol=1e-3)


# ---- (2) fixed point + (8) energy ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS_EMU)
@pytest.mark.parametrize("den", ["REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE", "REBLUR_SPECULAR"])
def test_reblur_fixed_point(request, pkg, api, backend, den):
    if backend == "emulated" and den != "REBLUR_DIFFUSE_SPECULAR":
        pytest.skip("emulated kernels: one variant is enough here (the bit-exact emulated-vs-oracle test covers all)")
    b = get(request, backend)
    w, h = size_for(backend, (96, 64), (48, 32))
    d = api.Denoiser[den]
    hz = pkg.harness.Harness(b, [d], w, h)
    fr = util.flat_frame(pkg, w, h)
    st = {d: api.ReblurSettings()}
    for f in range(4):
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
        for key, src in (("out_diff", "diff"), ("out_spec", "spec")):
            if (key == "out_diff" and "DIFFUSE" not in den) or (key == "out_spec" and "SPECULAR" not in den):
                continue
            out = hz.output(key)
            assert util.max_ulp_f16(out, fr[src]) <= 1, (f, key)


@pytest.mark.parametrize("backend", BACKENDS)
def test_reblur_energy_and_noise_reduction(request, pkg, api, backend):
    b = get(request, backend)
    w, h = 192, 128
    D = api.Denoiser
    d = D.REBLUR_DIFFUSE_SPECULAR
    hz = pkg.harness.Harness(b, [d], w, h)
    rng = np.random.default_rng(7)
    st = {d: api.ReblurSettings()}
    ins, outs = [], []
    for f in range(12):
        fr = util.flat_frame(pkg, w, h, rng=rng, sigma=0.5)
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
        ins.append(fr["diff"][..., 0].astype(np.float64))
        outs.append(hz.output("out_diff")[..., 0].astype(np.float64))
    true_y = 0.25 * 0.6 + 0.5 * 0.5 + 0.25 * 0.4
    inner = (slice(16, -16), slice(16, -16))
    # mean luminance preserved (weights are normalised), noise strongly reduced
    assert abs(outs[-1][inner].mean() / true_y - 1) < 0.02
    assert outs[-1][inner].std() < 0.1 * ins[-1][inner].std()


# ---- (4) out-of-range isolation + (5) sky -----------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS_EMU)
def test_garbage_isolation(request, pkg, api, backend):
    """NaN ("GARBAGE = sqrt(-1)", Shaders/Shared.hlsli:150) outside rectSize or beyond denoisingRange never reaches an output
    (USE_DRS_STRESS_TEST / USE_INF_STRESS_TEST, Shaders/Shared.hlsli:31-32)."""
    b = get(request, backend)
    res_w, res_h = size_for(backend, (128, 96), (64, 48))
    w, h = res_w - 24, res_h - 16  # DRS: rect < resource
    D = api.Denoiser
    dens = [D.REBLUR_DIFFUSE_SPECULAR, D.SIGMA_SHADOW_TRANSLUCENCY]
    scene = pkg.synth.Scene(w, h, dolly=0.02)
    results = []
    for garbage in (False, True):
        hz = pkg.harness.Harness(b, dens, res_w, res_h)
        st = util.default_settings(api, scene, dens)
        for f in range(3):
            fr = scene.frame(f)
            sky = fr["viewz"] > 1e4
            big = {}
            for key in ("viewz", "mv", "normal_roughness", "diff", "spec", "penumbra", "translucency"):
                a = fr[key]
                full = np.zeros((res_h, res_w) + a.shape[2:], dtype=a.dtype)
                if garbage and a.dtype in (np.float16, np.float32):
                    full[...] = np.nan
                elif garbage:
                    full[...] = np.iinfo(a.dtype).max
                a = a.copy()
                if garbage and key in ("mv", "diff", "spec", "penumbra"):
                    a[sky] = np.nan
                full[:h, :w] = a
                big[key] = full
            big["confidence"] = fr["confidence"]
            cs = scene.common_settings(api, fr, f, reset=(f == 0))
            for k in ("resourceSize", "resourceSizePrev"):
                getattr(cs, k)[0], getattr(cs, k)[1] = res_w, res_h
            hz.frame(cs, hz.upload(big), st)
        results.append({k: hz.fetch(v).copy() for k, v in hz.outputs.items()})
    for key, bpt in (("out_diff", 8), ("out_spec", 8), ("out_shadow", 4)):
        a = results[0][key].reshape(res_h, res_w, bpt)[:h, :w]
        g = results[1][key].reshape(res_h, res_w, bpt)[:h, :w]
        assert np.array_equal(a, g), key
        if bpt == 8:
            assert np.isfinite(np.ascontiguousarray(g).view(np.float16)).all()


@pytest.mark.parametrize("backend", BACKENDS)
def test_all_sky(request, pkg, api, backend):
    b = get(request, backend)
    w, h = 64, 48
    D = api.Denoiser
    dens = [D.REBLUR_DIFFUSE_SPECULAR, D.SIGMA_SHADOW_TRANSLUCENCY]
    hz = pkg.harness.Harness(b, dens, w, h)
    fr = util.flat_frame(pkg, w, h, z=1e5)
    st = {D.REBLUR_DIFFUSE_SPECULAR: api.ReblurSettings(), D.SIGMA_SHADOW_TRANSLUCENCY: api.SigmaSettings(lightDirection=[0, 1, 0])}
    for f in range(2):
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
    for key in ("out_diff", "out_spec", "out_shadow"):
        assert not hz.fetch(hz.outputs[key]).any()
    assert (hz.pool("REBLUR::Tiles")[: (h + 15) // 16, : (w + 15) // 16] == 1).all()
    assert not hz.pool("REBLUR::History").any()


# ---- (6) split screen --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
def test_split_screen(request, pkg, api, backend):
    b = get(request, backend)
    w, h = 96, 64
    D = api.Denoiser
    dens = [D.REBLUR_DIFFUSE_SPECULAR, D.SIGMA_SHADOW_TRANSLUCENCY, D.REFERENCE]
    scene = pkg.synth.Scene(w, h)
    hz = pkg.harness.Harness(b, dens, w, h)
    st = util.default_settings(api, scene, dens)
    for f in range(2):
        fr = scene.frame(f)
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        cs.splitScreen = 0.5
        planes = hz.upload(fr)
        hz.frame(cs, planes, st)
    left = slice(0, w // 2)
    assert np.array_equal(hz.output("out_diff")[:, left], fr["diff"][:, left])
    assert np.array_equal(hz.output("out_spec")[:, left], fr["spec"][:, left])
    assert not np.array_equal(hz.output("out_diff")[:, w // 2:], fr["diff"][:, w // 2:])
    sig = hz.fetch(planes["signal"]).view(np.float16).reshape(h, w, 4)
    assert np.array_equal(sig[:, left], fr["signal"][:, left])
    lit_in = fr["penumbra"][:, left] >= 65504
    sh = hz.output("out_shadow", np.uint8)[:, left, 0]
    hit = fr["viewz"][:, left] < 1e4
    assert np.array_equal(sh[hit] == 255, lit_in[hit])


# ---- (7) disocclusion ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
def test_disocclusion_resets_history(request, pkg, api, backend):
    b = get(request, backend)
    w, h = 96, 64
    D = api.Denoiser
    d = D.REBLUR_DIFFUSE
    hz = pkg.harness.Harness(b, [d], w, h)
    st = {d: api.ReblurSettings(maxAccumulatedFrameNum=30, maxFastAccumulatedFrameNum=30)}  # fast == slow disables clamping
    fr = util.flat_frame(pkg, w, h, z=5.0)
    for f in range(6):
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
    cur = 6 & 1 ^ 1  # data1 ping-pong written by frame 5
    a_before = hz.pool("REBLUR::Data1_B" if cur else "REBLUR::Data1_A").view(np.uint16)[:h, :w] & 0xFF
    assert (a_before == 5 * 4).all()  # 5 accumulated frames, stored in quarter frames
    fr2 = util.flat_frame(pkg, w, h, z=5.0)
    fr2["viewz"][16:48, 32:64] = 4.0  # a depth step far beyond disocclusionThreshold * frustum size, zero motion
    hz.frame(util.static_common(api, w, h, 6), hz.upload(fr2), st)
    a_after = hz.pool("REBLUR::Data1_A").view(np.uint16)[:h, :w] & 0xFF
    assert (a_after[20:44, 36:60] == 0).all()  # history discarded inside the uncovered region
    assert (a_after[:12, :] == 6 * 4).all()    # untouched far away


# ---- (9) SIGMA -------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS_EMU)
def test_sigma_lit_and_umbra(request, pkg, api, backend):
    b = get(request, backend)
    w, h = size_for(backend, (96, 64), (48, 32))
    D = api.Denoiser
    d = D.SIGMA_SHADOW_TRANSLUCENCY
    st = {d: api.SigmaSettings(lightDirection=[0, 0, -1])}
    # no occluder anywhere => shadow 1
    hz = pkg.harness.Harness(b, [d], w, h)
    fr = util.flat_frame(pkg, w, h)
    for f in range(3):
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
    assert (hz.output("out_shadow", np.uint8) == 255).all()
    # full occluder at distance 0 with a translucency tint => shadow 0, tint preserved (sqrt encoded)
    hz = pkg.harness.Harness(b, [d], w, h)
    fr["penumbra"][...] = 0
    fr["translucency"][...] = (0, 230, 150, 80)
    for f in range(3):
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
    out = hz.output("out_shadow", np.uint8)
    assert (out[..., 0] == 0).all()
    want = np.floor(np.sqrt(np.array([230, 150, 80]) / 255.0) * 255 + 0.5)
    assert np.abs(out[..., 1:].astype(np.int32) - want).max() <= 1


# ---- (10) determinism ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
def test_determinism(request, pkg, api, backend):
    b = get(request, backend)
    w, h = 160, 96
    D = api.Denoiser
    dens = [D.REBLUR_DIFFUSE_SPECULAR, D.SIGMA_SHADOW_TRANSLUCENCY]
    scene = pkg.synth.Scene(w, h, dolly=0.02)
    runs = []
    for rep in range(2):
        hz = pkg.harness.Harness(b, dens, w, h)
        if backend == "oracle":
            b.lib.orc_set_threads(hz.nrd.handle, 1 if rep == 0 else 5)  # row striping must not change a bit
        st = util.default_settings(api, scene, dens)
        for f in range(4):
            fr = scene.frame(f)
            hz.frame(scene.common_settings(api, fr, f, reset=(f == 0)), hz.upload(fr), st)
        runs.append(hz)
    assert util.compare_all(runs[0], runs[1], exact=True) == []


def test_oracle_worker_pool(pkg, api, oracle):
    """oracle/orc_core.cpp pool_run (round 6): rows are handed out in small chunks to persistent workers. More threads than rows, thread
    counts that change between frames and two instances driven from two host threads at once must all give the planes of one thread"""
    import threading

    w, h = 96, 80
    D = api.Denoiser
    dens = [D.REBLUR_DIFFUSE_SPECULAR, D.RELAX_DIFFUSE]
    scene = pkg.synth.Scene(w, h, dolly=0.02)
    frames = [scene.frame(f) for f in range(3)]

    def run(threads_per_frame, out, key):
        hz = pkg.harness.Harness(oracle, dens, w, h)
        st = util.default_settings(api, scene, dens)
        for f, fr in enumerate(frames):
            oracle.lib.orc_set_threads(hz.nrd.handle, threads_per_frame[f])
            hz.frame(scene.common_settings(api, fr, f, reset=(f == 0)), hz.upload(fr), st)
        out[key] = hz

    res = {}
    run([1, 1, 1], res, "one")
    run([300, 3, 64], res, "many")
    assert util.compare_all(res["one"], res["many"], exact=True) == []
    ts = [threading.Thread(target=run, args=([7, 16, 2], res, "a")), threading.Thread(target=run, args=([5, 1, 33], res, "b"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert util.compare_all(res["one"], res["a"], exact=True) == [] and util.compare_all(res["one"], res["b"], exact=True) == []


# ---- RELAX: fixed point (linear RGB + world-space hit distance in, same out) -------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS_EMU)
def test_relax_fixed_point(request, pkg, api, backend):
    b = get(request, backend)
    w, h = size_for(backend, (96, 64), (48, 32))
    d = api.Denoiser.RELAX_DIFFUSE_SPECULAR
    hz = pkg.harness.Harness(b, [d], w, h)
    fr = util.flat_frame(pkg, w, h)
    for key, rgb, hit in (("diff", (0.6, 0.5, 0.4), 1.5), ("spec", (0.3, 0.4, 0.5), 7.0)):
        fr[key] = np.broadcast_to(np.array(rgb + (hit,), dtype=np.float16), (h, w, 4)).copy()
    st = {d: api.RelaxSettings()}
    for f in range(4):
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
        for key, src in (("out_diff", "diff"), ("out_spec", "spec")):
            # RGB -> YCoCg -> RGB in fp16 planes: a few ULP on the colour channels, hit distance exact to 1 ULP
            assert util.max_ulp_f16(hz.output(key)[..., :3], fr[src][..., :3]) <= 4, (f, key)
            assert util.max_ulp_f16(hz.output(key)[..., 3], fr[src][..., 3]) <= 1, (f, key)
    names = [x["name"] for x in hz.nrd.dispatches([int(d)])]
    assert names[:4] == ["RELAX::ClassifyTiles", "RELAX::PrePass", "RELAX::TemporalAccumulation", "RELAX::HistoryFix"]
    assert len(names) == 4 + 5 and names[-1] == "RELAX::Atrous4"  # atrousIterationNum = 5 (Source/NRDSample.cpp:1642 range 2..8)


# ---- OCCLUSION variants: the normalised hit distance alone, R16_UNORM in / out (Source/NRDSample.cpp:488-501, :2934-2937) ----
@pytest.mark.parametrize("backend", BACKENDS)
def test_occlusion_fixed_point_and_denoising(request, pkg, api, backend):
    b = get(request, backend)
    w, h = 96, 64
    d = api.Denoiser.REBLUR_DIFFUSE_SPECULAR_OCCLUSION
    hz = pkg.harness.Harness(b, [d], w, h)
    fr = util.flat_frame(pkg, w, h)
    fr["diff_hitdist"] = np.full((h, w), 40000, dtype=np.uint16)
    fr["spec_hitdist"] = np.full((h, w), 12345, dtype=np.uint16)
    st = {d: api.ReblurSettings()}
    for f in range(3):
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
        for key, src in (("out_diff_hitdist", "diff_hitdist"), ("out_spec_hitdist", "spec_hitdist")):
            out = hz.fetch(hz.outputs[key]).view(np.uint16).reshape(h, w).astype(np.int32)
            assert np.abs(out - fr[src].astype(np.int32)).max() <= 32, (f, key)  # internal planes are fp16: 1 ULP = 32 LSB
    # noisy AO gets denoised, mean preserved
    hz = pkg.harness.Harness(b, [d], w, h)
    rng = np.random.default_rng(3)
    for f in range(8):
        fr["diff_hitdist"] = np.clip(rng.normal(30000, 9000, (h, w)), 0, 65535).astype(np.uint16)
        fr["spec_hitdist"] = fr["diff_hitdist"]
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
    out = hz.fetch(hz.outputs["out_diff_hitdist"]).view(np.uint16).reshape(h, w).astype(np.float64)
    assert abs(out[8:-8, 8:-8].mean() / 30000 - 1) < 0.03 and out[8:-8, 8:-8].std() < 0.25 * 9000


# ---- SH variants: SH1 is filtered with exactly the weights of SH0 (Source/NRDSample.cpp:464-476) ---------------------------
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("den", ["REBLUR_DIFFUSE_SPECULAR_SH", "RELAX_DIFFUSE_SPECULAR_SH"])
def test_sh_fixed_point_and_linearity(request, pkg, api, backend, den):
    b = get(request, backend)
    w, h = 96, 64
    d = api.Denoiser[den]
    relax = den.startswith("RELAX")
    st = {d: api.RelaxSettings() if relax else api.ReblurSettings()}
    fr = util.flat_frame(pkg, w, h, rng=np.random.default_rng(9))
    if relax:
        for key, hit in (("diff", 1.5), ("spec", 7.0)):
            fr[key][..., 3] = hit
    # SH1 proportional to SH0's first three channels: since both see the same weights, the output keeps the proportion
    k = 0.5
    fr["diff_sh1"] = (fr["diff"].astype(np.float32) * k).astype(np.float16)
    fr["spec_sh1"] = (fr["spec"].astype(np.float32) * k).astype(np.float16)
    hz = pkg.harness.Harness(b, [d], w, h)
    for f in range(3):
        hz.frame(util.static_common(api, w, h, f, reset=(f == 0)), hz.upload(fr), st)
    if not relax:  # REBLUR keeps YCoCg in SH0, so SH1 = k * SH0 must survive every pass (up to fp16 rounding of both planes)
        for a, b1 in (("out_diff", "out_diff_sh1"), ("out_spec", "out_spec_sh1")):
            sh0 = hz.output(a).astype(np.float32)[8:-8, 8:-8, 0]
            sh1 = hz.output(b1).astype(np.float32)[8:-8, 8:-8, 0]
            assert np.abs(sh1 / np.maximum(sh0, 1e-3) - k).max() < 0.01
    else:  # RELAX outputs linear RGB in SH0 while SH1 stays in its own space: just require a denoised, finite SH1
        sh1 = hz.output("out_diff_sh1").astype(np.float32)
        assert np.isfinite(sh1).all() and sh1[8:-8, 8:-8, 0].std() < 0.5 * fr["diff_sh1"].astype(np.float32)[8:-8, 8:-8, 0].std()


def test_directional_occlusion_fixed_point_and_layout(pkg, api, oracle):
    """REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION (Source/NRDSample.cpp:488-491): one {direction * h, h} texel in and out. A constant
    texel on a static flat plane is a fixed point of the whole pipeline; the split screen returns the noisy texel."""
    D = api.Denoiser
    w, h = 48, 32
    dens = [D.REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION]
    hz = pkg.harness.Harness(oracle, dens, w, h)
    names = None
    texel = np.array([0.1, -0.2, 0.3, 0.5], np.float16)
    for f in range(4):
        fr = util.flat_frame(pkg, w, h)
        fr["diff_dirocc"] = np.broadcast_to(texel, (h, w, 4)).copy()
        cs = util.static_common(api, w, h, frame_index=f, reset=(f == 0))
        hz.frame(cs, hz.upload(fr), {dens[0]: api.ReblurSettings()})
        assert util.max_ulp_f16(hz.output("out_diff_dirocc"), fr["diff_dirocc"]) <= 2, f
        names = [x["name"] for x in hz.nrd.dispatches([int(dens[0])])]
    assert names[:3] == ["REBLUR::ClassifyTiles", "REBLUR::PrepareInputs", "REBLUR::PrePass"]  # the split into SH0 / SH1 halves
    # split screen: left half shows the (noisy) input texel
    rng = np.random.default_rng(0)
    fr = util.flat_frame(pkg, w, h)
    fr["diff_dirocc"] = (np.broadcast_to(texel, (h, w, 4)) * (1 + 0.3 * rng.standard_normal((h, w, 1)))).astype(np.float16)
    cs = util.static_common(api, w, h, frame_index=4)
    cs.splitScreen = 0.5
    hz.frame(cs, hz.upload(fr), {dens[0]: api.ReblurSettings()})
    out = hz.output("out_diff_dirocc")
    assert np.array_equal(out[:, : w // 2].view(np.uint16), fr["diff_dirocc"][:, : w // 2].view(np.uint16))
    assert not np.array_equal(out[:, w // 2:].view(np.uint16), fr["diff_dirocc"][:, w // 2:].view(np.uint16))


def test_baseline_config1_reference_256(pkg, api, oracle, emulated):
    """BASELINE.json configs[0] as SURVEY.md 8d states it: REFERENCE, 256x256, analytic gradient x (1 + 0.5 U(-1, 1)), 64 frames,
    maxAccumulatedFrameNum 1024, CLEAR_AND_RESTART at frame 32 - on the CPU oracle and on the emulated kernels (no GPU)."""
    D = api.Denoiser
    w = h = 256
    x = (np.arange(w) + 0.5) / w
    y = ((np.arange(h) + 0.5) / h)[:, None]
    clean = np.stack([np.broadcast_to(x, (h, w)), np.broadcast_to(y, (h, w)), np.full((h, w), 0.5), np.ones((h, w))], -1)
    st = {D.REFERENCE: api.ReferenceSettings(maxAccumulatedFrameNum=1024)}
    outs = {}
    for name, b in (("oracle", oracle), ("emulated", emulated)):
        hz = pkg.harness.Harness(b, [D.REFERENCE], w, h)
        rng = np.random.default_rng(0x9E3779B9 & 0xFFFF)
        errs = []
        for f in range(64):
            noisy = clean.copy()
            noisy[..., :3] *= 1.0 + 0.5 * rng.uniform(-1, 1, (h, w, 1))
            planes = hz.upload({"signal": noisy.astype(np.float16)})
            cs = util.static_common(api, w, h, f, reset=(f == 0 or f == 32))
            hz.frame(cs, planes, st)
            out = hz.fetch(planes["signal"]).view(np.float16).reshape(h, w, 4).astype(np.float64)
            errs.append(((out[..., :3] - clean[..., :3]) ** 2).mean())
            if f == 32:  # restart: the output is the (fp16) input of this frame
                assert np.array_equal(out.astype(np.float16), noisy.astype(np.float16))
        outs[name] = (errs, out)
        # running mean: error variance ~ 1 / (frames since restart)
        assert errs[31] < errs[0] / 20 and errs[63] < errs[32] / 20 and errs[33] > 5 * errs[31]
    assert np.array_equal(outs["oracle"][1], outs["emulated"][1])
