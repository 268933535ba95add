"""The UNMODIFIED product kernel sources (nrd-sample_amd/csrc/*.hip + nrdhip.cpp), compiled for the host against
tests/hip_emu, must match the oracle bit for bit on every output AND every pool plane - this checks the kernels themselves
(indexing, LDS tiles, barriers, pass wiring) in the GPU-less container; the -m gpu tests then only have to establish that
gfx950 code generation keeps the same arithmetic."""
import pytest

import util


@pytest.mark.parametrize("dens", [["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW_TRANSLUCENCY", "REFERENCE"], ["REBLUR_DIFFUSE"],
                                  ["REBLUR_SPECULAR", "SIGMA_SHADOW"], ["RELAX_DIFFUSE_SPECULAR"], ["RELAX_DIFFUSE"], ["RELAX_SPECULAR"],
                                  ["REBLUR_DIFFUSE_SPECULAR_OCCLUSION"], ["REBLUR_DIFFUSE_OCCLUSION"], ["REBLUR_SPECULAR_OCCLUSION"],
                                  ["REBLUR_DIFFUSE_SPECULAR_SH"], ["REBLUR_SPECULAR_SH"], ["RELAX_DIFFUSE_SPECULAR_SH"], ["RELAX_DIFFUSE_SH"],
                                  ["REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION"]])
def test_emulated_kernels_bit_exact(pkg, api, oracle, emulated, dens):
    w, h = 72, 40  # not a multiple of 16: exercises partial tiles
    scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR")
    dd = [api.Denoiser[x] for x in dens]
    st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1)
    frames = 2 if ("_SH" in dens[0] or "OCCLUSION" in dens[0]) else 3  # keep the (slow) emulated runs within the CPU-suite budget
    ho = util.run_frames(api, pkg.harness, oracle, scene, dd, frames, settings=st)
    he = util.run_frames(api, pkg.harness, emulated, scene, dd, frames, settings=st)
    assert util.compare_all(ho, he, exact=True) == []


def roughness_table_run(pkg, api, backend, dens, w, h, frames=3):
    """frames whose settings CHANGE from one to the next in everything the roughness table of a frame is made from (round 6:
    ReblurParams::roughLut - hit distance parameters C / D, through the hit distance factor) and in what is evaluated beside it per pixel
    (roughnessFraction, lobeAngleFraction): a table left over from the frame before would show in every specular pixel"""
    D = api.Denoiser
    scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR")
    dd = [D[x] for x in dens]
    hz = pkg.harness.Harness(backend, dd, w, h)
    for f in range(frames):
        fr = scene.frame(f)
        st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1)
        for d, s in st.items():
            if isinstance(s, api.ReblurSettings):
                s.hitDistanceParameters = api.ReblurHitDistanceParameters(A=3.0 + f, B=0.1, C=20.0 - 6.0 * f, D=-25.0 + 8.0 * f)
                s.roughnessFraction, s.lobeAngleFraction = 0.15 + 0.1 * f, 0.15 + 0.2 * f
            elif isinstance(s, api.RelaxSettings):
                s.roughnessFraction, s.lobeAngleFraction, s.specularLobeAngleSlack = 0.15 + 0.1 * f, 0.5 - 0.15 * f, 0.15 + 0.1 * f
        hz.frame(scene.common_settings(api, fr, f, reset=(f == 0)), hz.upload(fr), st)
    return hz


@pytest.mark.parametrize("dens", [["REBLUR_DIFFUSE_SPECULAR"], ["REBLUR_SPECULAR_SH"], ["RELAX_DIFFUSE_SPECULAR"]])
def test_roughness_table_follows_the_settings_of_every_frame(pkg, api, oracle, emulated, dens):
    ho = roughness_table_run(pkg, api, oracle, dens, 72, 40)
    he = roughness_table_run(pkg, api, emulated, dens, 72, 40)
    assert util.compare_all(ho, he, exact=True) == []


def test_emulated_dispatch_lists_match(pkg, api, oracle, emulated):
    D = api.Denoiser
    dens = [D.REBLUR_DIFFUSE_SPECULAR, D.SIGMA_SHADOW_TRANSLUCENCY, D.REFERENCE]
    scene = pkg.synth.Scene(64, 48)
    lists = []
    for b in (oracle, emulated):
        hz = pkg.harness.Harness(b, dens, 64, 48)
        fr = scene.frame(0)
        hz.nrd.set_common_settings(scene.common_settings(api, fr, 0, reset=True))
        lists.append(hz.nrd.dispatches([int(d) for d in dens]))
    assert [x["name"] for x in lists[0]] == [x["name"] for x in lists[1]]
    assert len(lists[0]) == 6 + 5 + 1  # (REBLUR: PrePass + TemporalAccumulation are one dispatch)
    for a, b in zip(*lists):
        assert a["written"] == b["written"] and a["read"] == b["read"] and a["halo_rows"] == b["halo_rows"]
        assert abs(a["bytes_per_pixel"] - b["bytes_per_pixel"]) < 1e-4
    total = sum(x["bytes_per_pixel"] for x in lists[1] if x["name"].startswith("REBLUR"))
    # REBLUR_DIFFUSE_SPECULAR algorithmic bytes / pixel / frame: SURVEY.md 8d estimated ~352 with 8-byte guides (this build's guide
    # texel since round 3): 408 of round 2 + 16 (HistoryFix writes tap texels) + 16 (Blur writes them; its guide read is gone)
    # - 6 x 8 (the 8-byte guide texel) = 392 with separate passes; the fused PrePass + TemporalAccumulation dispatch neither writes nor
    # reads Tmp1 (-32), fetches the guide once (-8) and does not read the hit tracker back (-2): 350
    assert 346 < total < 354


@pytest.mark.parametrize("dens", [["REBLUR_DIFFUSE_SPECULAR"], ["REBLUR_DIFFUSE"], ["REBLUR_SPECULAR"]])
def test_fused_prepass_is_bit_identical_to_separate_passes(pkg, api, oracle, emulated, dens):
    """REBLUR radiance flavours run PrePass + TemporalAccumulation as ONE dispatch (csrc/nrd_reblur.hip spatial_pixel<..., FUSED>) unless
    the instance is created with NRDHIP_FLAG_SEPARATE_PASSES: outputs and every pool plane except the then untouched Tmp1 must be
    bit-identical, in the kernels and in the oracle, which mirrors both dispatch lists"""
    w, h = 72, 56
    scene = pkg.synth.Scene(w, h, dolly=0.04)
    dd = [api.Denoiser[x] for x in dens]
    st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1, enableAntiFirefly=True)
    hs = {}
    for tag, (b, sep) in dict(of=(oracle, False), os=(oracle, True), ef=(emulated, False), es=(emulated, True)).items():
        hs[tag] = pkg.harness.Harness(b, dd, w, h, separate_passes=sep)
    names = lambda hz: [x["name"].split("::")[1] for x in hz.nrd.dispatches([int(d) for d in dd])]
    for f in range(3):
        fr = scene.frame(f)
        for hz in hs.values():
            hz.frame(scene.common_settings(api, fr, f, reset=(f == 0)), hz.upload(fr), st)
        assert names(hs["ef"])[:3] == ["ClassifyTiles", "PrePassTemporalAccumulation", "HistoryFix"] == names(hs["of"])[:3]
        assert names(hs["es"])[:4] == ["ClassifyTiles", "PrePass", "TemporalAccumulation", "HistoryFix"] == names(hs["os"])[:4]
        assert util.compare_all(hs["of"], hs["ef"], exact=True) == []
        assert util.compare_all(hs["os"], hs["es"], exact=True) == []
        assert [x for x in util.compare_all(hs["ef"], hs["es"], exact=True) if not x[0].endswith("::Tmp1")] == []
    import numpy as np
    assert not np.asarray(hs["ef"].pool("REBLUR::Tmp1")).any() and np.asarray(hs["es"].pool("REBLUR::Tmp1")).any()  # the fused frame never touches it


def test_hw_transcendentals_flavour_places_agree(pkg, api, oracle_hwt, emulated_hwt):
    """-DNRD_HW_TRANSCENDENTALS=1 on both sides: the kernels' transcendental builtins are IEEE 1 / x, sqrtf, exp2f under the host emulation -
    exactly what liboracle_hwt.so computes at those places - so the two must agree bit for bit: every weight-class site is switched on BOTH
    sides and nowhere else (on the GPU the instructions are 1 ULP: tests/test_hw_transcendentals.py holds that build to the 1-ULP-fp16 bar)"""
    w, h = 72, 56
    for den, kw in (("REBLUR_DIFFUSE_SPECULAR", {}), ("REBLUR_DIFFUSE_SPECULAR", dict(ortho=True)), ("RELAX_DIFFUSE_SPECULAR_SH", {}), ("REBLUR_DIFFUSE_SPECULAR_SH", {})):
        scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if den.startswith("RELAX") else "REBLUR", **kw)
        dd = [api.Denoiser[den]]
        st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1)
        ho = util.run_frames(api, pkg.harness, oracle_hwt, scene, dd, 3, settings=st)
        he = util.run_frames(api, pkg.harness, emulated_hwt, scene, dd, 3, settings=st)
        assert util.compare_all(ho, he, exact=True) == []


@pytest.mark.parametrize("dens", [["REBLUR_DIFFUSE_SPECULAR"], ["RELAX_DIFFUSE_SPECULAR_SH"]])
def test_emulated_kernels_sky_tiles(pkg, api, oracle, emulated, dens):
    """a frame tall enough that whole tiles are sky: HistoryFix / TemporalStabilization skip their staging on tiles ClassifyTiles
    marked (the Tiles mask), every thread taking the per-pixel sky path - outputs and pools must still equal the oracle's, which
    knows no such shortcut; camera rolled too, so that the sky tiles form a column band instead of rows"""
    import numpy as np

    for roll, (w, h) in ((0.0, (40, 104)), (90.0, (104, 40))):  # 3 x 7 / 7 x 3 tiles, partial ones on both axes
        scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR", roll_deg=roll)
        dd = [api.Denoiser[x] for x in dens]
        st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1)
        ho = util.run_frames(api, pkg.harness, oracle, scene, dd, 2, settings=st)
        he = util.run_frames(api, pkg.harness, emulated, scene, dd, 2, settings=st)
        assert util.compare_all(ho, he, exact=True) == []
        tiles = np.asarray(he.pool(("RELAX" if dens[0].startswith("RELAX") else "REBLUR") + "::Tiles"))
        assert tiles.max() == 1 and tiles.min() == 0  # the run had sky tiles and geometry tiles


@pytest.mark.parametrize("dens,kw", [(["REBLUR_DIFFUSE_SPECULAR"], dict(enableAntiFirefly=True)), (["RELAX_DIFFUSE_SPECULAR"], {}),
                                     (["REBLUR_DIFFUSE_SPECULAR_SH"], {}), (["REBLUR_DIFFUSE_SPECULAR_OCCLUSION"], {}),
                                     (["RELAX_DIFFUSE_SPECULAR_SH"], {})])
def test_nobody_reads_what_sky_tiles_leave_unwritten(pkg, api, emulated, dens, kw):
    """PrePass, TemporalAccumulation and PostBlur write nothing in tiles without geometry (csrc/nrd_reblur.hip k_spatial): whatever
    their planes - and every other internal plane - hold at pixels beyond the denoising range must not matter. Two instances of the
    emulated kernels run the same frames; in one of them every internal plane except the guides is overwritten, before every frame,
    with random fp16 bit patterns at the pixels that are beyond the range in the previous AND the current frame (a texel that
    had geometry a frame ago holds history the reprojection is entitled to). Every output of every frame must be bit-identical: no
    pass consumes such a texel other than through a test of the guide that selects it out, or with a weight of exactly 0.
    The PERMANENT planes get finite patterns only (they are cleared before a denoiser's first frame and only ever written with clamped
    fp16 values, which is what allows "weight 0 x texel" there); the TRANSIENT planes - one arena aliased across denoisers, so anything
    may be left in it - get every pattern, NaN and Inf included: a texel of theirs may only ever be SELECTED out."""
    import numpy as np

    rng = np.random.default_rng(7)
    for roll, (w, h) in ((0.0, (40, 104)), (90.0, (104, 40))):
        scene = pkg.synth.Scene(w, h, dolly=0.06, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR", roll_deg=roll)
        dd = [api.Denoiser[x] for x in dens]
        st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1, **kw)
        if dens[0].startswith("RELAX"):
            st = {dd[0]: api.RelaxSettings(enableAntiFirefly=True, minMaterialForDiffuse=0, minMaterialForSpecular=1)}
        ha = pkg.harness.Harness(emulated, dd, w, h)
        hb = pkg.harness.Harness(emulated, dd, w, h)
        prev_sky = None
        poisoned = 0
        for f in range(5):
            fr = scene.frame(f)
            cs = scene.common_settings(api, fr, f, reset=(f == 0))
            sky = np.abs(fr["viewz"].astype(np.float32) * float(cs.viewZScale)) > float(cs.denoisingRange)
            if prev_sky is not None:
                both = sky & prev_sky
                for pool in (0, 1):
                    for p in hb.nrd.pools[pool]:
                        if "Guide" in p["name"] or p["height"] != h or p["width"] != w:
                            continue
                        words = p["bpt"] // 2
                        if words == 0:
                            continue
                        arr = p["buf"].view(np.uint16).reshape(h, -1)[:, : w * words].reshape(h, w, words)
                        junk = rng.integers(0, 1 << 16, size=arr.shape, dtype=np.uint16)
                        if pool == 0:
                            junk = np.where((junk & 0x7c00) == 0x7c00, junk & np.uint16(0xbbff), junk)  # no Inf / NaN: finite fp16 only
                        else:
                            junk[..., 0] = np.where(junk[..., 0] % 3 == 0, np.uint16(0x7e00), np.where(junk[..., 0] % 3 == 1, np.uint16(0xfc00), junk[..., 0]))  # NaN, -Inf
                        arr[both] = junk[both]
                        poisoned += int(both.sum())
            prev_sky = sky
            ha.frame(cs, ha.upload(fr), st)
            hb.frame(cs, hb.upload(fr), st)
            for key in ha.outputs:
                assert np.array_equal(ha.fetch(ha.outputs[key]), hb.fetch(hb.outputs[key])), (roll, f, key)
        assert poisoned > 0
        tiles = np.asarray(hb.pool(("RELAX" if dens[0].startswith("RELAX") else "REBLUR") + "::Tiles"))
        assert tiles.max() == 1 and tiles.min() == 0


@pytest.mark.parametrize("dens", [["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW"], ["RELAX_DIFFUSE_SPECULAR"]])
def test_no_hit_depth_patterns_are_sky(pkg, api, oracle, emulated, dens):
    """what a renderer writes into IN_VIEWZ where a ray hit nothing must not matter - 1e5 (the sample, Shared.hlsli:141), +-Inf, a
    NaN, or a plane cleared with 0xFF bytes (a NaN whose bit pattern used to WRAP in the guide's depth rounding and come back as geometry
    at depth ~0). ClassifyTiles stores one canonical finite depth for all of them, so every plane and every output is bit-identical to
    the run with plain far-away depths - in the oracle and in the kernels"""
    import numpy as np

    w, h = 72, 88
    scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR")
    dd = [api.Denoiser[x] for x in dens]
    st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1)
    patterns = np.array([0xFFFFFFFF, 0x7F800000, 0xFF800000, 0x7FC00000, 0xFFFFFE00, 0x7F7FFFFF], dtype=np.uint32)

    def mangle(f, fr):
        cs = scene.common_settings(api, fr, f)
        sky = np.abs(fr["viewz"].astype(np.float32) * float(cs.viewZScale)) > float(cs.denoisingRange)
        assert sky.any() and not sky.all()
        z = fr["viewz"].astype(np.float32).copy().view(np.uint32)
        yy, xx = np.nonzero(sky)
        z[yy, xx] = patterns[(xx + 3 * yy + f) % len(patterns)]
        fr["viewz"] = z.view(np.float32)

    plain = util.run_frames(api, pkg.harness, oracle, scene, dd, 3, settings=st)
    odd = util.run_frames(api, pkg.harness, oracle, scene, dd, 3, settings=st, frame_hook=mangle)
    emu = util.run_frames(api, pkg.harness, emulated, scene, dd, 3, settings=st, frame_hook=mangle)
    assert util.compare_all(plain, odd, exact=True) == []
    assert util.compare_all(odd, emu, exact=True) == []
    for key in ("out_diff", "out_spec"):
        assert np.isfinite(odd.output(key).astype(np.float32)).all()
