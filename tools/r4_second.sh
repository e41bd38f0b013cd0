#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gather_gpu.py tests/test_batch_gpu.py -m gpu -x -q > gpurun_out/r4_second_tests.log 2>&1
tail -5 gpurun_out/r4_second_tests.log
python bench.py --workload cfg5 --steps 10 --warmup 3 > gpurun_out/r4_cfg5.log 2>&1; tail -c 1000 gpurun_out/r4_cfg5.log | cut -c1-700
SNOUT_BENCH_NCCL1=1 SNOUT_BENCH_FAKE_WORLD=8 python bench.py --workload cfg5 --steps 10 --warmup 3 > gpurun_out/r4_cfg5_fake8.log 2>&1; tail -c 1000 gpurun_out/r4_cfg5_fake8.log| cut -c1-700
WL=cfg5 PER=2 STEPS=6 LIST=16 bash tools/r4_tl.sh
