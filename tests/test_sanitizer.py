"""Sanitizer leg: the product kernel sources compiled for the host with -fsanitize=address,undefined
-fno-sanitize-recover (tests/_emu/libnrdhip_emu_san.so, __graft_entry__.build_emulated(sanitize=True)) run a cross-section of the
emulated-kernel tests in a child process that has the ASan runtime preloaded. The host emulation is the one place where an
out-of-bounds tap, footprint or LDS staging index faults loudly instead of reading a neighbour's bytes (on the GPU the planes are
padded by nothing and a stray read is silent): every denoiser family, PrepareInputs with a checkerboard, DRS with rectSize <
resourceSize changing between frames, a frame narrower and shorter than one 16x16 tile, sky tiles (incl. the run that overwrites
what they leave unwritten), the orthographic flavour, anti-firefly on SH texels, and the 2-band row tiler."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

SELECTION = [
    "tests/test_kernels_emulated.py::test_emulated_kernels_bit_exact[dens0]",   # REBLUR_DIFFUSE_SPECULAR + SIGMA_SHADOW_TRANSLUCENCY + REFERENCE
    "tests/test_kernels_emulated.py::test_emulated_kernels_bit_exact[dens3]",   # RELAX_DIFFUSE_SPECULAR (A-trous LDS windows)
    "tests/test_kernels_emulated.py::test_emulated_kernels_bit_exact[dens9]",   # REBLUR_DIFFUSE_SPECULAR_SH
    "tests/test_kernels_emulated.py::test_emulated_kernels_bit_exact[dens6]",   # REBLUR_DIFFUSE_SPECULAR_OCCLUSION
    "tests/test_kernels_emulated.py::test_emulated_kernels_bit_exact[dens10]",  # RELAX_DIFFUSE_SPECULAR_SH
    "tests/test_kernels_emulated.py::test_emulated_kernels_bit_exact[dens12]",  # REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION
    "tests/test_kernels_emulated.py::test_emulated_kernels_sky_tiles[dens0]",
    "tests/test_kernels_emulated.py::test_emulated_kernels_sky_tiles[dens1]",
    "tests/test_kernels_emulated.py::test_nobody_reads_what_sky_tiles_leave_unwritten[dens0-kw0]",
    "tests/test_ortho.py::test_ortho_emulated_bit_exact[REBLUR_DIFFUSE_SPECULAR+SIGMA_SHADOW_TRANSLUCENCY+REFERENCE]",
    "tests/test_settings_variants.py::test_variants_emulated_bit_exact[antifirefly_relax_sh]",
    "tests/test_prepare_inputs.py::test_prepare_inputs_emulated_bit_exact[dens0-WHITE-AREA_5X5]",  # checkerboard + 5x5 hit-distance reconstruction
    "tests/test_prepare_inputs.py::test_prepare_inputs_emulated_bit_exact[dens1-BLACK-None]",
    "tests/test_settings_variants.py::test_variants_emulated_bit_exact[drs_reblur_sigma]",
    "tests/test_sanitizer.py::test_tiny_frames_emulated",
    "tests/test_tiler_gloo.py::test_row_tiling_emulated_kernels_bit_identical",
]


def test_tiny_frames_emulated(pkg, api, oracle, emulated):
    """frames narrower / shorter than one 16x16 tile, and a 1-pixel-wide one: every clamp and window bound of the kernels"""
    import util

    D = api.Denoiser
    for (w, h), dens in (((10, 9), [D.REBLUR_DIFFUSE_SPECULAR, D.SIGMA_SHADOW_TRANSLUCENCY]), ((1, 23), [D.REBLUR_DIFFUSE_SPECULAR]),
                         ((21, 3), [D.RELAX_DIFFUSE_SPECULAR]), ((17, 17), [D.REBLUR_DIFFUSE_SPECULAR_SH])):
        scene = pkg.synth.Scene(w, h, dolly=0.04, denoiser="RELAX" if dens[0].name.startswith("RELAX") else "REBLUR")
        st = util.default_settings(api, scene, dens, minMaterialForDiffuse=0, minMaterialForSpecular=1)
        ho = util.run_frames(api, pkg.harness, oracle, scene, dens, 3, settings=st)
        he = util.run_frames(api, pkg.harness, emulated, scene, dens, 3, settings=st)
        assert util.compare_all(ho, he, exact=True) == [], (w, h)


@pytest.mark.skipif(os.environ.get("NRD_EMU_SANITIZE") == "1", reason="this IS the sanitized child process")
def test_emulated_kernels_under_address_and_ub_sanitizer():
    if not os.path.exists(graft.SANITIZER_RUNTIME):
        pytest.skip("no ASan runtime in this image")
    graft.build_emulated(sanitize=True)  # built here, once, before the (parallel) child processes want it
    env = dict(os.environ)
    env.update(NRD_EMU_SANITIZE="1", LD_PRELOAD=graft.SANITIZER_RUNTIME,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-n", "6"] + SELECTION
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "failed" not in r.stdout, tail


def test_sanitizer_sees_through_the_emulation_fibers(tmp_path):
    """the leg above is only worth something if ASan still catches a stray access made from a fiber stack (the emulation runs every HIP
    thread of a barrier kernel as a ucontext fiber and announces the stack switches): tests/host/asan_fiber_probe.cpp runs clean
    without arguments and must die with a heap-buffer-overflow report when told to write one element past a plane"""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang) or not os.path.exists(graft.SANITIZER_RUNTIME):
        pytest.skip("no clang / ASan runtime in this image")
    exe = str(tmp_path / "asan_fiber_probe")
    subprocess.run([clang, "-O1", "-std=c++17", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + os.path.join(ROOT, "tests", "hip_emu"), os.path.join(ROOT, "tests", "host", "asan_fiber_probe.cpp"), "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    ok = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and "sum 32640" in ok.stdout, ok.stdout + ok.stderr
    bad = subprocess.run([exe, "fault"], env=env, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "heap-buffer-overflow" in bad.stderr, bad.stderr[-2000:]
