#!/bin/bash
# Dev tool (gpurun): SQ counters of the 802.15.4 channelizer pfb_spec<16> (cfg #4, one segment at a time), two --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pfb16; mkdir -p $O
B="python3 $R/bench.py"
run() { timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/sq_$1 -- $B --workload cfg4 --steps 2 --warmup 1 --no-cpu --sync > $O/sq_$1.log 2>&1; }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
run c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pfb_spec<16" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc): print("%-28s %14.0f per launch (%d launches)" % (k, acc[k][0] / acc[k][1], acc[k][1]))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
