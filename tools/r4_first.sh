#!/bin/bash
# round 4, first GPU pass: new batch / on-device tests, then cfg5 and cfg3 timings
mkdir -p gpurun_out
python -m pytest tests/test_batch_gpu.py tests/test_wideband_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/r4_first_tests.log 2>&1
tail -5 gpurun_out/r4_first_tests.log
python bench.py --workload cfg5 --steps 10 --warmup 3 > gpurun_out/r4_cfg5.log 2>&1; tail -c 1500 gpurun_out/r4_cfg5.log
python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu > gpurun_out/r4_cfg3.log 2>&1; tail -c 1200 gpurun_out/r4_cfg3.log
python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu > gpurun_out/r4_cfg4.log 2>&1; tail -c 800 gpurun_out/r4_cfg4.log
