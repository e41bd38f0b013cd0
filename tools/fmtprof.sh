cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for w in zigbee1 cfg3 cfg4; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fmt_$w -- python3 $R/bench.py --workload $w --format sc8 --no-cpu --sync --steps 10 > $O/prof_fmt_$w.log 2>&1
f=$(ls $O/prof_fmt_$w/*/*kernel_stats.csv | head -1); echo "== $w"; grep snout $f | cut -d, -f1-4 | head -12
done
timeout 300 python3 $R/bench.py --workload cfg3 --format sc8 --no-cpu | cut -c1-330
timeout 300 python3 $R/bench.py --workload cfg3 --no-cpu | cut -c1-330
