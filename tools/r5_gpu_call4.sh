#!/bin/bash
# round 5, GPU call 4: the tile order table (FrameConsts::tileTable) - parity of the in-tree library (table on), then A/B of table vs computed
# tile on the headline (with the no-sky leg), the rolled camera and config 4
mkdir -p gpurun_out/r5d
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_graph.py tests/test_history_rows.py -m gpu -q -x -k "not 8k and not 4k" --durations=5 > gpurun_out/r5d/pytest_parity.txt 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r5d/pytest_parity.txt
timeout 900 python tools/ab.py --rounds 3 --workload reblur_ds_4k --full-coverage t0_computed t1_table > gpurun_out/r5d/ab_headline.txt 2>&1
tail -4 gpurun_out/r5d/ab_headline.txt
timeout 600 python tools/ab.py --rounds 2 --workload reblur_ds_4k --bench-args "--roll 90" t0_computed t1_table > gpurun_out/r5d/ab_roll90.txt 2>&1
tail -3 gpurun_out/r5d/ab_roll90.txt
timeout 600 python tools/ab.py --rounds 2 --workload relax_ds_sh_4k t0_computed t1_table > gpurun_out/r5d/ab_relax_sh.txt 2>&1
tail -3 gpurun_out/r5d/ab_relax_sh.txt
timeout 600 python tools/ab.py --rounds 2 --workload reblur_ds_sigma_1440p t0_computed t1_table > gpurun_out/r5d/ab_config3.txt 2>&1
tail -3 gpurun_out/r5d/ab_config3.txt
