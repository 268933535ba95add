#!/bin/bash
# One bounded GPU-box session, most important first: default bench line of HEAD, 2-rank plumbing dry runs of the row-tiled bench
# (BASELINE config 5's 8K strong-scaling mode; gloo, both ranks on the one GPU: numbers meaningless, the N > 1 code path is what is
# exercised) with the Python tiler and with the C++ tiler, GPU parity of the newest cases + smoke, then an optional A/B of
# prebuilt library variants (_variants/*.so, see tools/ab_variants.sh).
out=gpurun_out/${1:-final}
mkdir -p $out
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?" > $out/status.txt
for tiler in python native; do
  NRD_BENCH_DRYRUN_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --tiler $tiler > $out/bench_dryrun2_$tiler.json 2> $out/bench_dryrun2_$tiler.err
  echo "dryrun2 8K $tiler rc=$?" >> $out/status.txt
done
timeout 900 python -m pytest tests/test_tiler_gloo.py tests/test_cpp_harness.py tests/test_abi.py -m gpu -q > $out/pytest_new_cases.log 2>&1
echo "pytest rc=$?" >> $out/status.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
echo "smoke rc=$?" >> $out/status.txt
if [ -d _variants ] && [ -n "$AB" ]; then
    ROUNDS=${ROUNDS:-2} bash tools/ab_variants.sh $AB > $out/ab.log 2>&1
    echo "ab rc=$?" >> $out/status.txt
fi
cat $out/status.txt; cat $out/bench_default.json; tail -2 $out/bench_dryrun2_python.json; tail -3 $out/bench_dryrun2_python.err; tail -2 $out/bench_dryrun2_native.json; tail -3 $out/bench_dryrun2_native.err; tail -15 $out/pytest_new_cases.log; tail -3 $out/smoke.log; cat $out/ab.log 2>/dev/null
