#!/bin/bash
# One bounded GPU-box session, most important first: default bench line of HEAD, 2-rank plumbing dry run of the tiled bench
# (gloo, both ranks on the one GPU: numbers meaningless, the N > 1 code path is what is exercised), GPU parity of the newest
# cases + smoke, then an A/B of prebuilt library variants (_variants/*.so, see tools/ab_variants.sh).
out=gpurun_out/${1:-final}
mkdir -p $out
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?" > $out/status.txt
NRD_BENCH_DRYRUN_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_dryrun2.json 2> $out/bench_dryrun2.err
echo "dryrun2 rc=$?" >> $out/status.txt
timeout 240 python -m pytest tests/test_settings_variants.py -m gpu -q -k "camera_attached or strand_reblur" > $out/pytest_new_cases.log 2>&1
echo "pytest rc=$?" >> $out/status.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
echo "smoke rc=$?" >> $out/status.txt
if [ -d _variants ] && [ -n "$AB" ]; then
    ROUNDS=${ROUNDS:-2} bash tools/ab_variants.sh $AB > $out/ab.log 2>&1
    echo "ab rc=$?" >> $out/status.txt
fi
cat $out/status.txt; cat $out/bench_default.json; tail -3 $out/bench_dryrun2.json; tail -3 $out/pytest_new_cases.log; cat $out/ab.log 2>/dev/null
