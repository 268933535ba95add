"""VALU instructions of one gfx950 kernel attributed to the source line of a chosen INLINING DEPTH (tools/isa_lines.py attributes to the innermost frame,
which piles everything onto fma_() / lerpf()): depth 0 = the line inside the kernel function itself, 1 = the line inside the function the kernel
calls, ... The .loc comments of an -S build with line tables carry the whole inlined-at chain.
usage: python tools/isa_frames.py <file.hip> <kernel substring> <depth> [--under LINE] [extra hipcc flags ...]
  --under LINE: only instructions whose frame at depth-1 is LINE (e.g. the call site of ta_pixel inside spatial_pixel)"""
import collections, re, subprocess, sys

args = sys.argv[1:]
src, pat, depth = args[0], args[1], int(args[2])
extra = args[3:]
under = None
if "--under" in extra:
    i = extra.index("--under")
    under = int(extra[i + 1])
    del extra[i:i + 2]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
       "-fno-slp-vectorize", "-gline-tables-only", "-x", "hip", "--cuda-device-only", "-S", src, "-o", "/tmp/isa_frames.s"] + extra
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, cwd="nrd-sample_amd/csrc")
s = open("/tmp/isa_frames.s").read()
names = [m.group(1) for m in re.finditer(r"^(_ZN[^\n:]*):", s, re.M) if pat in m.group(1)]
name = names[0]
body = s[s.index(name + ":"):]
body = body[:body.index(".Lfunc_end")]
cur, per, slow = None, collections.Counter(), collections.Counter()
FAST = re.compile(r"v_(fma_f32|fmac_f32|fmaak_f32|fmamk_f32|mul_f32|add_f32|sub_f32|subrev_f32|mov_b32|and_b32|or_b32|xor_b32|add_u32|sub_u32|subrev_u32|add_co_u32|not_b32)\b")
for line in body.split("\n"):
    m = re.match(r"\s+\.loc\s+\d+\s+\d+.*?;\s*(.*)$", line)
    if m:
        frames = re.findall(r"([\w./]+):(\d+):\d+", m.group(1))  # innermost first
        frames = [(f.split("/")[-1], int(l)) for f, l in frames][::-1]  # outermost first
        cur = frames
        continue
    m = re.match(r"\s+(v_\w+)", line)
    if m and cur:
        if under is not None and (len(cur) <= depth - 1 or depth < 1 or cur[depth - 1][1] != under):
            continue
        key = cur[depth] if len(cur) > depth else cur[-1]
        per[key] += 1
        if not FAST.match(m.group(1)):
            slow[key] += 1
cache = {}
def text(f, l):
    if f not in cache:
        try:
            cache[f] = open("nrd-sample_amd/csrc/" + f).read().split("\n")
        except OSError:
            cache[f] = []
    t = cache[f]
    return t[l - 1].strip()[:120] if 0 < l <= len(t) else ""
total = sum(per.values())
print(name, "VALU", total, "slow-class", sum(slow.values()))
for (f, l), n in sorted(per.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print("%5d (%4d slow) %4.1f%%  %s:%d  %s" % (n, slow[(f, l)], 100.0 * n / total, f, l, text(f, l)))
