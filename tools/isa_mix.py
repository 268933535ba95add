"""Static VALU instruction mix of the gfx950 code of the kernels in a translation unit, by issue class (tools/ubench/valu_rate.hip:
v_fma / v_fmac / v_fmaak / v_fmamk / v_mul / v_add / v_sub _f32 issue in 2.4-2.6 cycles per wave64 at >= 4 waves per SIMD, nearly
everything else - conversions, compares, v_med3, v_cndmask, integer and bit-field ops, v_fma_mix, v_floor - in ~4).
usage: python tools/isa_mix.py <file.hip> [kernel substring] [extra hipcc flags ...]
The spatial kernels are straight-line code (one basic block per tap), so the static mix is the dynamic mix of a wave with geometry."""
import re, subprocess, sys, collections

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
       "-fno-slp-vectorize", "-x", "hip", "--cuda-device-only", "-S", src, "-o", "/tmp/isa_mix.s"] + extra
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, cwd="nrd-sample_amd/csrc")
s = open("/tmp/isa_mix.s").read()
FAST = re.compile(r"^v_(fma_f32|fmac_f32|fmaak_f32|fmamk_f32|mul_f32|add_f32|sub_f32|subrev_f32|mac_f32)(_e32|_e64)?$")
for m in re.finditer(r"^(_ZN[^\n:]*):.*?; Occupancy: \d+", s, re.S | re.M):
    name = m.group(1)
    if pat not in name:
        continue
    body = m.group(0)
    code = body.split(".Lfunc_end")[0]
    ops = re.findall(r"^\s+(v_[a-z0-9_]+)", code, re.M)
    fast = sum(1 for o in ops if FAST.match(o))
    slow = len(ops) - fast
    vg = re.search(r"; NumVgprs: (\d+)", body).group(1)
    sc = re.search(r"; ScratchSize: (\d+)", body).group(1)
    oc = re.search(r"; Occupancy: (\d+)", body).group(1)
    nl = len(re.findall(r"^\s+(buffer_load|global_load)", code, re.M))
    short = re.sub(r"_ZN6nrdhip\d*(_GLOBAL__N_1|5ortho12_GLOBAL__N_1)?", "", name).replace("NS_12ReblurParamsE", "")
    print("%-44s vgpr %3s scratch %3s occ %s loads %3d | VALU %4d = fast %4d + slow %4d -> %.0f cycles at 2.5 / 4.0 (mean %.2f per instruction)" % (
        short[:44], vg, sc, oc, nl, len(ops), fast, slow, fast * 2.5 + slow * 4.0, (fast * 2.5 + slow * 4.0) / max(len(ops), 1)))
    if len(sys.argv) > 2 and "--top" in extra:
        pass
    top = collections.Counter(o for o in ops if not FAST.match(o)).most_common(14)
    print("     slow ops: " + ", ".join("%s %d" % (o.replace("_e32", "").replace("_e64", ""), n) for o, n in top))
