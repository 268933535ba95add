#!/bin/bash
# tools/kstat_dir.sh <source dir> <file.hip> <kernel-substring> [extra flags]: registers / VALU / branches of the kernels of a TU in a variant source directory
D=$1; F=$2; PAT=$3; shift; shift; shift
cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-slp-vectorize "$@" -x hip --cuda-device-only -S $F -o /tmp/kstat_$$.s 2>/dev/null
python3 - "$PAT" /tmp/kstat_$$.s <<'PY'
import re,sys
pat=sys.argv[1]
s=open(sys.argv[2]).read()
for m in re.finditer(r'^(_ZN[^\n:]*):.*?; Occupancy: \d+', s, re.S|re.M):
    name=m.group(1)
    if pat not in name: continue
    body=m.group(0)
    code=body.split('.Lfunc_end')[0]
    vg=re.search(r'; NumVgprs: (\d+)',body); sc=re.search(r'; ScratchSize: (\d+)',body); oc=re.search(r'; Occupancy: (\d+)',body)
    nv=len(re.findall(r'^\s+v_',code,re.M)); nb=len(re.findall(r'^\s+s_cbranch',code,re.M)); nt=len(re.findall(r'^\s+v_(rcp|sqrt|rsq|exp|log)_f32',code,re.M))
    short=re.sub(r'_ZN6nrdhip\d*(_GLOBAL__N_1|5ortho12_GLOBAL__N_1)?','',name).replace('NS_12ReblurParamsE','')
    print('%-52s vgpr %3s scratch %4s occ %s  valu %4d (trans %2d) br %2d' % (short[:52], vg.group(1), sc.group(1), oc.group(1), nv, nt, nb))
PY
rm -f /tmp/kstat_$$.s
