#!/bin/bash
# round 5, GPU call 8: kernel arguments of the tile look-up and of the first vector loads pinned at kernel entry (one trip to the argument segment) A/B
mkdir -p gpurun_out/r5h
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "not 8k and not 4k and not 1440p" > gpurun_out/r5h/pytest_parity.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r5h/pytest_parity.txt
timeout 900 python tools/ab.py --rounds 3 --workload reblur_ds_4k --full-coverage pin0 pin1 > gpurun_out/r5h/ab_headline.txt 2>&1
tail -3 gpurun_out/r5h/ab_headline.txt
timeout 600 python tools/ab.py --rounds 2 --workload reblur_d_1080p pin0 pin1 > gpurun_out/r5h/ab_1080p.txt 2>&1
tail -3 gpurun_out/r5h/ab_1080p.txt
timeout 600 python tools/ab.py --rounds 2 --workload relax_ds_sh_4k pin0 pin1 > gpurun_out/r5h/ab_relax_sh.txt 2>&1
tail -3 gpurun_out/r5h/ab_relax_sh.txt
