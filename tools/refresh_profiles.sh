#!/bin/bash
# Run on the MI355X box (through gpurun): every rocprofv3 pass behind profiles/r1_*.
# Outputs land in gpurun_out/prof_*; tools/collect_profiles_all.py copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 20 --no-cpu > $O/prof_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench_sync -- python3 $R/bench.py --steps 20 --no-cpu --sync > $O/prof_bench_sync.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/prof_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/prof_pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wide -- python3 $R/tools/quick_wide.py > $O/prof_wide.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_zbbig -- python3 $R/tools/zb_big.py > $O/prof_zbbig.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_zb_pmc_fetch -- python3 $R/tools/zb_big.py > $O/prof_zb_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_zb_pmc_write -- python3 $R/tools/zb_big.py > $O/prof_zb_pmc_write.log 2>&1
timeout 600 python3 $R/bench.py --steps 20 > $O/bench_plain.log 2>&1
tail -n 1 $O/prof_bench.log | cut -c1-400; tail -n 1 $O/prof_bench_sync.log | cut -c1-200; tail -n 3 $O/prof_wide.log; tail -n 1 $O/prof_zbbig.log; tail -n 1 $O/bench_plain.log | cut -c1-300
