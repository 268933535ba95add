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
    # history counters: unknown identifier / null pointer are INVALID_ARGUMENT; a fresh denoiser reports "no history yet"
    import ctypes
    st = api.HistoryState(9, 9, 9, 9)
    assert b.get_history_state(nrd.handle, int(D.REFERENCE), ctypes.byref(st)) == 0
    assert (st.frame_counter, st.frames_since_reset, st.history_valid) == (0, 0, 0)
    assert b.get_history_state(nrd.handle, 4242, ctypes.byref(st)) == int(api.Result.INVALID_ARGUMENT)
    assert b.set_history_state(nrd.handle, int(D.REFERENCE), None) == int(api.Result.INVALID_ARGUMENT)
    if which == "emulated":  # NRDHIP_FLAG_GRAPH bookkeeping exists in the product library only
        assert nrd.graph_stats() == dict(replayed=0, instantiated=0, direct=0)
    nrd.destroy()


def _common(api, w, h):
    cs = api.CommonSettings()
    cs.rectSize[0] = cs.resourceSize[0] = cs.rectSizePrev[0] = cs.resourceSizePrev[0] = w
    cs.rectSize[1] = cs.resourceSize[1] = cs.rectSizePrev[1] = cs.resourceSizePrev[1] = h
    cs.viewToClipMatrix[0] = cs.viewToClipMatrix[5] = cs.viewToClipMatrix[11] = 1.0
    cs.viewToClipMatrixPrev[0] = cs.viewToClipMatrixPrev[5] = cs.viewToClipMatrixPrev[11] = 1.0
    for i in (0, 5, 10, 15):
        cs.worldToViewMatrix[i] = cs.worldToViewMatrixPrev[i] = 1.0
    return cs


def test_boundary_checks_of_the_product_host_code(api, emulated):
    """VERDICT r1 item 8 / ADVICE r1 on the product's host code (its host-emulated build runs here): the denoiser-kind query behind
    nrd::Integration::SetDenoiserSettings, the band query, slot planes smaller than the rect are refused, a non-zero rectOrigin is
    refused, unbind_all drops stale slot pointers, an unknown device ordinal is refused at creation."""
    import ctypes as C
    import numpy as np

    D = api.Denoiser
    nrd = api.Integration(emulated)
    assert nrd.recreate([(1, D.REBLUR_DIFFUSE), (2, D.RELAX_SPECULAR), (3, D.SIGMA_SHADOW), (4, D.REFERENCE)], 64, 48) == api.Result.SUCCESS
    kind = C.c_uint32()
    for ident, want in ((1, 0), (2, 1), (3, 2), (4, 3)):
        assert emulated.denoiser_kind(nrd.handle, ident, C.byref(kind)) == 0 and kind.value == want
    assert emulated.denoiser_kind(nrd.handle, 99, C.byref(kind)) == int(api.Result.INVALID_ARGUMENT)
    band = (C.c_int32 * 5)()
    assert emulated.get_band(nrd.handle, band) == 0 and list(band) == [48, 0, 0, 48, 48]
    nrd.set_common_settings(_common(api, 64, 48))
    sig = np.zeros((48, 64 * 8), np.uint8)
    small = np.zeros((24, 64 * 8), np.uint8)
    nrd.set_resource(api.ResourceType.IN_SIGNAL, sig, api.Format.RGBA16_SFLOAT, width=64, height=48)
    nrd.set_resource(api.ResourceType.OUT_SIGNAL, small, api.Format.RGBA16_SFLOAT, width=64, height=24)  # half the rect's rows
    with pytest.raises(api.NrdError) as e:
        nrd.denoise([4])
    assert e.value.code == api.Result.INVALID_ARGUMENT and "smaller than the rect" in str(e.value)
    nrd.set_resource(api.ResourceType.OUT_SIGNAL, sig, api.Format.RGBA16_SFLOAT, width=64, height=48)
    nrd.denoise([4])  # in place, fine
    info = api.PlaneInfo()
    assert emulated.slot_info(nrd.handle, int(api.ResourceType.IN_SIGNAL), C.byref(info)) == 0 and info.ptr == sig.ctypes.data
    emulated.lib.nrdhip_unbind_all(C.c_void_p(nrd.handle.value))
    assert emulated.slot_info(nrd.handle, int(api.ResourceType.IN_SIGNAL), C.byref(info)) == 0 and not info.ptr
    with pytest.raises(api.NrdError):
        nrd.denoise([4])  # nothing bound any more
    cs = _common(api, 64, 48)
    cs.rectOrigin[0] = 8
    nrd.set_common_settings(cs)
    nrd.set_resource(api.ResourceType.IN_SIGNAL, sig, api.Format.RGBA16_SFLOAT, width=64, height=48)
    nrd.set_resource(api.ResourceType.OUT_SIGNAL, sig, api.Format.RGBA16_SFLOAT, width=64, height=48)
    with pytest.raises(api.NrdError) as e:
        nrd.denoise([4])
    assert "rectOrigin" in str(e.value)
    nrd.destroy()
    # device ordinal: the emulated runtime has exactly one device
    arr = (api.DenoiserDesc * 1)(api.DenoiserDesc(4, int(D.REFERENCE)))
    h = C.c_void_p()
    desc = api.CreateDesc(arr, 1, 64, 48, 0, 0, 0, 6, 0, api.FLAG_EXTERNAL_POOLS)  # device 5
    assert emulated.create(C.byref(desc), C.byref(h)) == int(api.Result.INVALID_ARGUMENT)
    desc = api.CreateDesc(arr, 1, 64, 48, 0, 0, 0, 1, 0, api.FLAG_EXTERNAL_POOLS)  # device 0
    assert emulated.create(C.byref(desc), C.byref(h)) == 0
    emulated.destroy(h)


def test_empty_row_window_still_does_the_bookkeeping(api, emulated):
    """nrdhip_denoise_rows with a window outside the owned rows and PART_FIRST (but not PART_LAST) used to be a no-op -
    the CLEAR_AND_RESTART clear of the permanent pool, which rides on the first part, was then never issued"""
    import numpy as np

    D = api.Denoiser
    nrd = api.Integration(emulated)
    assert nrd.recreate([(4, D.REFERENCE)], 32, 32) == api.Result.SUCCESS
    hist = nrd.pool_plane("REFERENCE::History")["buf"]
    hist[:] = 0x55  # stale history
    cs = _common(api, 32, 32)
    cs.accumulationMode = int(api.AccumulationMode.CLEAR_AND_RESTART)
    nrd.set_common_settings(cs)
    sig = np.zeros((32, 32 * 8), np.uint8)
    nrd.set_resource(api.ResourceType.IN_SIGNAL, sig, api.Format.RGBA16_SFLOAT, width=32, height=32)
    nrd.set_resource(api.ResourceType.OUT_SIGNAL, sig, api.Format.RGBA16_SFLOAT, width=32, height=32)
    nrd.denoise_rows([4], 0, 64, 16, part=nrd.PART_FIRST)  # rows 64..79: outside the 32 rows this instance owns
    assert not hist.any(), "the clear of the permanent pool must not depend on the first part having rows to compute"
    nrd.destroy()


def test_rccl_entry_points_do_not_take_the_library_down_without_a_gpu(pkg):
    """librccl is resolved lazily (dlopen at nrdhip_tiler_rccl_*): without a device the call reports FAILURE, nothing aborts"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    import subprocess
    import sys

    code = ("import ctypes as C; l = C.CDLL(%r); b = (C.c_uint8 * 128)(); r = l.nrdhip_tiler_rccl_unique_id(b); print('rc', r)" % pkg.HIP_LIB)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "rc 1" in r.stdout, r.stdout + r.stderr


def test_library_owned_transient_pool_is_one_aliased_arena(api, emulated):
    """with library-owned pools (the C++ path) the transient planes of all denoisers alias into ONE arena sized for the
    hungriest denoiser - the "aliasable" figure of Get*MemoryUsageInMb is real, and an all-denoiser instance like the sample's does
    not pay the sum"""
    import ctypes as C

    D = api.Denoiser

    def mem(dens, flags):
        arr = (api.DenoiserDesc * len(dens))(*[api.DenoiserDesc(int(d), int(d)) for d in dens])
        h = C.c_void_p()
        desc = api.CreateDesc(arr, len(dens), 256, 128, 0, 0, 0, 0, 0, flags)
        assert emulated.create(C.byref(desc), C.byref(h)) == 0
        out = (C.c_float * 3)()
        emulated.get_memory_mb(h, out)
        infos = []
        n = C.c_uint32()
        emulated.pool_size(h, 1, C.byref(n))
        for i in range(n.value):
            info = api.PlaneInfo()
            emulated.pool_info(h, 1, i, C.byref(info))
            infos.append((info.ptr, info.pitch_bytes * info.height))
        emulated.destroy(h)
        return list(out), infos

    one, _ = mem([D.REBLUR_DIFFUSE_SPECULAR], 0)
    two, infos = mem([D.REBLUR_DIFFUSE_SPECULAR, D.RELAX_DIFFUSE_SPECULAR, D.SIGMA_SHADOW], 0)
    relax, _ = mem([D.RELAX_DIFFUSE_SPECULAR], 0)
    assert abs(two[2] - max(one[2], relax[2])) < 1e-3  # aliasable = the hungriest denoiser, not the sum
    assert two[1] > one[1] + relax[1] - 1e-3           # persistent planes are never shared
    ext, _ = mem([D.REBLUR_DIFFUSE_SPECULAR, D.RELAX_DIFFUSE_SPECULAR, D.SIGMA_SHADOW], api.FLAG_EXTERNAL_POOLS)
    assert ext[2] > two[2] * 1.5                       # caller-owned pools are described plane by plane (the caller may alias them)
    ptrs = [p for p, _ in infos]
    assert len(set(ptrs)) < len(ptrs) and all(p for p in ptrs)  # planes of different denoisers share addresses
    lo, hi = min(ptrs), max(p + b for p, b in infos)
    assert hi - lo <= int(two[2] * 1048576) + 256
