#!/bin/bash
# Dev tool: SQ counters of the channelizer kernel (run through gpurun).  usage: tools/pfb_pmc.sh <variant> <tag> [proto]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$2; mkdir -p $O
export SNOUT_RX_LIB=$R/build/variants/libsnout_rx_$1.so
P=${3:-0}
run() { timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python3 $R/tools/pfb_ab.py --child --proto $P --samples 4e8 > $O/$1.log 2>&1; }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM"
run c "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
python3 - <<PY
import csv, glob, collections
for tag in "abc":
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "pfb_" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            for c, v in d.items():
                print(f"{k:60s} {c:28s} {v / cnt[(k, c)]:.4g} per dispatch ({cnt[(k, c)]} dispatches)")
PY
