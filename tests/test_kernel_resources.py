"""Register / scratch budget of the shipped gfx950 kernels, read from the code objects inside nrd-sample_amd/csrc/libnrdhip.so (no GPU
needed: hipcc cross-compiles). Guards what the measurements of DESIGN.md section 5 rest on: no kernel touches scratch memory (48 bytes
of it cost TemporalAccumulation 74 %, a struct select through 36 bytes cost SIGMA's stabilization its restructuring gain), and the
headline kernels keep the occupancy they were tuned at."""
import os
import re
import struct
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, tmp_path):
    """the gfx950 ELFs of every translation unit bundled into the library's .hip_fatbin section"""
    blob = str(tmp_path / "fatbin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + blob, lib, str(tmp_path / "unused.so")], check=True)
    data = open(blob, "rb").read()
    out, pos = [], data.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(MAGIC, pos + len(MAGIC))
    return out


def kernel_table(lib, tmp_path):
    table = {}
    for i, elf in enumerate(code_objects(lib, tmp_path)):
        path = str(tmp_path / ("co%d.elf" % i))
        open(path, "wb").write(elf)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True, check=True).stdout
        for block in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            get = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, block).group(1))
            agpr = int(re.match(r"\s*(\d+)", block).group(1))
            table[name] = dict(vgpr=get("vgpr_count"), agpr=agpr, scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"))
    return table


def waves_per_simd(k):
    regs = (k["vgpr"] + k["agpr"] + 7) // 8 * 8  # unified 512-entry file, granule 8
    return min(8, 512 // max(regs, 8))


@pytest.fixture(scope="module")
def kernels(pkg, tmp_path_factory):
    if not os.path.exists(os.path.join(LLVM, "llvm-readelf")) or not os.path.exists(pkg.HIP_LIB):
        pytest.skip("no llvm tools / no built product library")
    t = kernel_table(pkg.HIP_LIB, tmp_path_factory.mktemp("co"))
    assert len(t) > 60  # every flavour of every pass, perspective + orthographic
    return t


def test_no_kernel_uses_scratch(kernels):
    bad = {k: v["scratch"] for k, v in kernels.items() if v["scratch"]}
    assert not bad, bad


def test_headline_kernels_keep_their_occupancy(kernels):
    """REBLUR_DIFFUSE_SPECULAR radiance kernels, perspective flavour: waves per SIMD as tuned (profiles/r03_ab_pipeline_depth.txt,
    r03_ab_tap_texels.txt)"""
    want = {  # (mangled template arguments: k_spatial<VARIANT, MODE, HAS_DIFF, HAS_SPEC>, ...)
        "k_spatialILi0ELi0ELb1ELb1EE": 5,   # PrePass as its own dispatch (NRDHIP_FLAG_SEPARATE_PASSES), 2 taps in flight (frozen flavour: 6 waves)
        "k_spatialILi1ELi0ELb1ELb1EE": 5,   # Blur on tap texels, 6 taps in flight (frozen flavour: 8)
        "k_spatialILi2ELi0ELb1ELb1EE": 6,   # PostBlur, 2 taps in flight (frozen flavour: 3)
        "k_temporal_accumulationILb1ELb1ELb0ELb0EE": 4,
        "k_prepass_temporal_accumulationILb1ELb1EE": 4,  # the fused dispatch of record: 5 taps in flight, the reprojection half sets the registers
        "k_history_fixILb1ELb1ELb0EE": 7,
        "k_temporal_stabilizationILb1ELb1ELb0EE": 7,
        "k_classify_tiles": 8,
    }
    persp = {k: v for k, v in kernels.items() if "ortho" not in k}
    for frag, waves in want.items():
        hits = [(k, v) for k, v in persp.items() if frag in k]
        assert hits, frag
        for k, v in hits:
            assert waves_per_simd(v) >= waves, (k, v, waves_per_simd(v))
