#!/bin/bash
# Run on the MI355X box (through gpurun): every rocprofv3 pass behind profiles/r6_* (round 6).
# Outputs land in gpurun_out/r6prof/*; tools/collect_profiles_r6.py copies the summaries into profiles/.
# Every traced run is a profiling run of the STEADY STATE: --no-cpu (no CPU legs, no in-run one-lane checks), and
# tools/check_trace.py marks a directory whose dominant kernel's max exceeds 10 x its median (the collector refuses it).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6prof; mkdir -p $O
B="python3 $R/bench.py"
C="python3 $R/tools/check_trace.py"
# headline (cfg #3, 8e8 wideband samples): kernel stats, pipelined and one segment at a time
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- $B --steps 20 --no-cpu --no-others > $O/bench.log 2>&1
$C $O/bench pfb_spec btle_corr_planes
python3 $R/tools/timeline.py $O/bench --marker "pfb_spec<40" --last 6 --per 1 --skip 3 --list 6 > $O/bench_timeline.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_sync -- $B --steps 20 --no-cpu --no-others --sync > $O/bench_sync.log 2>&1
$C $O/bench_sync pfb_spec btle_corr_planes
# HBM traffic of the headline kernel: FETCH_SIZE and WRITE_SIZE in separate passes
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B --steps 2 --warmup 1 --no-cpu --no-others > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B --steps 2 --warmup 1 --no-cpu --no-others > $O/pmc_write.log 2>&1
# the other workloads, one at a time
for w in cfg2 cfg4 zigbee1; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- $B --workload $w --steps 10 --no-cpu --sync > $O/$w.log 2>&1
  $C $O/$w
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${w}_fetch -- $B --workload $w --steps 2 --warmup 1 --no-cpu --sync > $O/${w}_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${w}_write -- $B --workload $w --steps 2 --warmup 1 --no-cpu --sync > $O/${w}_write.log 2>&1
done
# cfg #5 (north_star's multi-GPU workload on this one GPU): the steady state only
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5 -- $B --workload cfg5 --steps 10 --warmup 3 --no-cpu > $O/cfg5.log 2>&1
$C $O/cfg5 pfb_spec zb_mm zb_walk zb_repair
python3 $R/tools/timeline.py $O/cfg5 --marker "pfb_spec<40" --last 6 --per 1 --skip 3 --list 9 > $O/cfg5_timeline.txt 2>&1
# cfg #4 pipelined (the frame repair and zb_walk beside the next segment's channelizer)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4_pipe -- $B --workload cfg4 --steps 10 --no-cpu > $O/cfg4_pipe.log 2>&1
$C $O/cfg4_pipe pfb_spec zb_mm zb_walk zb_repair
python3 $R/tools/timeline.py $O/cfg4_pipe --marker "pfb_spec<16" --last 6 --per 1 --skip 3 --list 9 > $O/cfg4_timeline.txt 2>&1
# rank 0's load of eight ranks rehearsed at world 1 (RCCL backend): step inflation of the headline workload and of cfg #5
bash $R/tools/r4_fake_world.sh > $O/fake_world.txt 2>&1
# SQ counters of the channelizer (three passes)
run() { timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/sq_$1 -- $B --steps 2 --warmup 1 --no-cpu --no-others > $O/sq_$1.log 2>&1; }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
run c "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
# per-wave cycle stamps of the headline channelizer (a -DSNOUT_MF_STAMPS build of pfb_spec.hip: tools/pfb_variants.sh spstamps:"-DSNOUT_MF_STAMPS")
[ -f $R/build/variants/libsnout_rx_spstamps.so ] && SNOUT_RX_LIB=$R/build/variants/libsnout_rx_spstamps.so timeout 600 python3 $R/tools/mf_stamps.py > $O/spec40_stamps.txt 2>&1
# the per-dispatch traces are large and not needed once the stats exist (gpurun copies back <= 64 MiB); of the counter
# passes only this library's kernels are kept (the captures' generation is thousands of torch dispatches)
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for f in $(find $O -name "*counter_collection.csv"); do (head -1 $f; grep "snout::" $f) > $f.tmp; mv $f.tmp $f; done
du -sh $O
# the unprofiled line the round is judged on, for comparison with the traced runs
( time timeout 1500 $B > $O/bench_plain.log 2> $O/bench_plain.err ) 2> $O/bench_plain.time
tail -n 1 $O/bench.log | cut -c1-300; tail -n 1 $O/bench_plain.log | cut -c1-300; cat $O/bench_plain.time; echo "traces that are not the steady state: $(cat $O/*/steady.txt | grep -c NOT_STEADY)"
