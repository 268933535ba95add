#!/bin/bash
# tools/kstat.sh <file.hip> [kernel-substring] [extra flags]: register / spill / branch summary of the gfx950 code of the kernels in a TU
F=$1; PAT=${2:-k_}; shift; shift
cd /root/repo/nrd-sample_amd/csrc
SLP="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden $SLP "$@" -x hip --cuda-device-only -S $F -o /tmp/kstat.s 2>/dev/null
python3 - "$PAT" <<'PY'
import re,sys
pat=sys.argv[1]
s=open('/tmp/kstat.s').read()
for m in re.finditer(r'^(_ZN[^\n:]*):.*?; Occupancy: \d+', s, re.S|re.M):
    name=m.group(1)
    if pat not in name: continue
    body=m.group(0)
    code=body  # (whole function: kernels with an early exit have several s_endpgm)
    vg=re.search(r'; NumVgprs: (\d+)',body); sc=re.search(r'; ScratchSize: (\d+)',body); oc=re.search(r'; Occupancy: (\d+)',body); sg=re.search(r'; TotalNumSgprs: (\d+)',body)
    nv=len(re.findall(r'^\s+v_',code,re.M)); nb=len(re.findall(r'^\s+s_cbranch',code,re.M)); nl=len(re.findall(r'^\s+global_load',code,re.M)); nw=len(re.findall(r'^\s+s_waitcnt vmcnt',code,re.M))
    short=re.sub(r'_ZN6nrdhip\d*(_GLOBAL__N_1|5ortho12_GLOBAL__N_1)?','',name).replace('NS_12ReblurParamsE','')
    print('%-48s vgpr %3s sgpr %3s scratch %4s occ %s  valu %4d br %2d loads %2d vmwaits %2d' % (short[:48], vg.group(1), sg.group(1), sc.group(1), oc.group(1), nv, nb, nl, nw))
PY
