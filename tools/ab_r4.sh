#!/bin/bash
# A/B of prebuilt library variants (_variants/<name>.so) on ONE box, alternating: usage tools/ab_r4.sh <rounds> v1 v2 ...
R=$1; shift
cp nrd-sample_amd/csrc/libnrdhip.so /tmp/keep.so
ROUNDS=$R bash tools/ab_variants.sh "$@"
cp /tmp/keep.so nrd-sample_amd/csrc/libnrdhip.so
