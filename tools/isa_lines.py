"""VALU instructions of one gfx950 kernel attributed to SOURCE LINES (innermost inlined frame, from the .loc directives of an -S build with
line tables): where a kernel's instructions come from, function by function - per-pixel set-up against the tap loops, software
reciprocals, decodes. usage: python tools/isa_lines.py <file.hip> <kernel substring> [extra hipcc flags ...]   (top 45 lines + totals by function)"""
import collections, re, subprocess, sys

src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
       "-fno-slp-vectorize", "-gline-tables-only", "-x", "hip", "--cuda-device-only", "-S", src, "-o", "/tmp/isa_lines.s"] + extra
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, cwd="nrd-sample_amd/csrc")
s = open("/tmp/isa_lines.s").read()
files = {int(m.group(1)): m.group(2) for m in re.finditer(r'^\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s, re.M)}
files.update({int(m.group(1)): m.group(2) for m in re.finditer(r'^\s+\.file\s+(\d+)\s+"([^"]+)"\s*$', s, re.M)})
names = [m.group(1) for m in re.finditer(r"^(_ZN[^\n:]*):", s, re.M) if pat in m.group(1)]
name = names[0]
body = s[s.index(name + ":"):]
body = body[:body.index(".Lfunc_end")]
cur, per = (0, 0), collections.Counter()
for line in body.split("\n"):
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if re.match(r"\s+v_", line):
        per[cur] += 1
lines_cache = {}
def text(f, l):
    path = files.get(f, "?")
    if path not in lines_cache:
        try:
            lines_cache[path] = open(path if path.startswith("/") else "nrd-sample_amd/csrc/" + path).read().split("\n")
        except OSError:
            lines_cache[path] = []
    t = lines_cache[path]
    return t[l - 1].strip()[:110] if 0 < l <= len(t) else ""
total = sum(per.values())
print(name, "VALU", total)
for (f, l), n in per.most_common(45):
    print("%5d %4.1f%%  %s:%d  %s" % (n, 100.0 * n / total, files.get(f, "?").split("/")[-1], l, text(f, l)))
