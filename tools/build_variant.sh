#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>": builds _variants/<name>.so from the current sources with extra flags
# (A/B timing on one GPU box with tools/ab_variants.sh). Objects go to a private directory so the in-tree library is untouched.
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/_variants/obj_$NAME
cd $ROOT/nrd-sample_amd/csrc
make -s -j16 OUT=$ROOT/_variants/$NAME.so OBJDIR=$ROOT/_variants/obj_$NAME EXTRA="$EXTRA" $ROOT/_variants/$NAME.so
