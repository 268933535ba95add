"""C-ABI library: loads without a GPU, exports every symbol include/nrdhip.h declares, struct sizes match the Python mirror,
and the host-side error behaviour mirrors the reference's nrd::Result convention (Source/NRDSample.cpp:958-959, 982-983)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nrdhip.h")).read()
    return sorted(set(re.findall(r"NRDHIP_API\s+[\w\s\*]+?\b(nrdhip_\w+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for must in ("nrdhip_create", "nrdhip_destroy", "nrdhip_set_common", "nrdhip_set_denoiser", "nrdhip_bind", "nrdhip_denoise"):
        assert must in syms


def test_library_exports_every_declared_symbol(pkg):
    assert os.path.exists(pkg.HIP_LIB), "libnrdhip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(pkg.HIP_LIB)
    for s in declared_symbols():
        assert hasattr(lib, s), "libnrdhip.so does not export %s" % s


def test_struct_sizes_match(pkg):
    pkg.hip_library_symbols()  # raises on mismatch
    import __graft_entry__ as graft
    graft.oracle_backend()


def test_library_desc_and_strings(pkg):
    lib = ctypes.CDLL(pkg.HIP_LIB)
    out = (ctypes.c_uint32 * 5)()
    assert lib.nrdhip_library_desc(out) == 0
    assert out[3] == 2 and out[4] == 1  # R10G10B10A2 normals, linear roughness (CMakeLists.txt:136-137)
    lib.nrdhip_denoiser_string.restype = ctypes.c_char_p
    assert lib.nrdhip_denoiser_string(18) == b"REFERENCE"
    assert lib.nrdhip_denoiser_string(6) == b"REBLUR_DIFFUSE_SPECULAR"


@pytest.mark.parametrize("which", ["oracle", "emulated"])
def test_error_convention(request, api, which):
    b = request.getfixturevalue(which)
    nrd = api.Integration(b)
    D = api.Denoiser
    # unknown denoiser and duplicate identifiers are reported, not raised (sample: "!= SUCCESS -> return false")
    assert nrd.recreate([(0, 200)], 64, 64) == api.Result.UNSUPPORTED  # not an nrd::Denoiser enumerator
    assert nrd.recreate([(7, D.REFERENCE), (7, D.SIGMA_SHADOW)], 64, 64) == api.Result.NON_UNIQUE_IDENTIFIER
    assert nrd.recreate([(int(D.REFERENCE), D.REFERENCE)], 64, 64) == api.Result.SUCCESS
    # Denoise before SetCommonSettings / with unknown identifier / with unbound slots fails with INVALID_ARGUMENT
    with pytest.raises(api.NrdError) as e:
        nrd.denoise([int(D.REFERENCE)])
    assert e.value.code == api.Result.INVALID_ARGUMENT
    cs = api.CommonSettings()
    cs.rectSize[0] = cs.rectSize[1] = cs.resourceSize[0] = cs.resourceSize[1] = 64
    cs.viewToClipMatrix[0] = cs.viewToClipMatrix[5] = cs.viewToClipMatrix[11] = 1.0
    cs.viewToClipMatrixPrev[0] = cs.viewToClipMatrixPrev[5] = cs.viewToClipMatrixPrev[11] = 1.0
    for i in (0, 5, 10, 15):
        cs.worldToViewMatrix[i] = cs.worldToViewMatrixPrev[i] = 1.0
    nrd.set_common_settings(cs)
    with pytest.raises(api.NrdError):
        nrd.denoise([12345])
    with pytest.raises(api.NrdError):
        nrd.denoise([int(D.REFERENCE)])  # IN_SIGNAL / OUT_SIGNAL not bound
    with pytest.raises(api.NrdError):
        nrd.set_denoiser_settings(int(D.REFERENCE), api.SigmaSettings())  # wrong settings struct for the denoiser
    # a slot bound with a format the backend does not read is refused, not misread
    import numpy as np
    sig = np.zeros((64, 64 * 8), np.uint8)
    nrd.set_resource(api.ResourceType.IN_SIGNAL, sig, api.Format.RGBA16_SFLOAT, width=64, height=64)
    nrd.set_resource(api.ResourceType.OUT_SIGNAL, sig, api.Format.RGBA32_SFLOAT, width=32, height=64)
    with pytest.raises(api.NrdError) as e:
        nrd.denoise([int(D.REFERENCE)])
    assert e.value.code == api.Result.INVALID_ARGUMENT and "format" in str(e.value)
    mem = nrd.memory_usage_mb()
    assert abs(mem["persistent"] - 64 * 64 * 16 / 1048576.0) < 1e-6
    nrd.destroy()
