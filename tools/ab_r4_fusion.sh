cp nrd-sample_amd/csrc/libnrdhip.so /tmp/keep.so
for r in 1 2 3; do
  cp _variants/head.so nrd-sample_amd/csrc/libnrdhip.so
  BENCH_ARGS="--separate-passes" ROUNDS=1 bash tools/ab_variants.sh head | sed 's/^head/separate/'
  ROUNDS=1 bash tools/ab_variants.sh head pd3 pd4 pd5
done
cp /tmp/keep.so nrd-sample_amd/csrc/libnrdhip.so
