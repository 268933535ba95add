#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>" [patch ...]: builds _variants/<name>.so from the current sources with extra flags and,
# optionally, with experiment patches applied (tools/variants/*.patch: shelved or timing-only code paths that do not live in the product
# sources) - A/B timing on one GPU box with tools/ab_variants.sh. Sources are copied to a private directory, so the in-tree library and
# sources are untouched.
set -e
NAME=$1; EXTRA=$2; shift; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/_variants/src_$NAME
rm -rf $SRC; mkdir -p $SRC
cp $ROOT/nrd-sample_amd/csrc/*.hip $ROOT/nrd-sample_amd/csrc/*.h $ROOT/nrd-sample_amd/csrc/*.cpp $ROOT/nrd-sample_amd/csrc/Makefile $SRC/
for P in "$@"; do (cd $SRC && patch -s -p3 < $ROOT/$P); done
cd $SRC
make -s -j16 OUT=$ROOT/_variants/$NAME.so OBJDIR=$SRC/_obj EXTRA="$EXTRA" $ROOT/_variants/$NAME.so
