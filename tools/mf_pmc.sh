#!/bin/bash
# Dev tool: SQ counters of the M = 40 channelizer kernel chosen by SNOUT_PFB_IMPL (run through gpurun).
#   tools/mf_pmc.sh <impl> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$2; mkdir -p $O
export SNOUT_PFB_IMPL=$1
run() { timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python3 $R/tools/pfb_ab.py --child --proto 0 --samples 4e8 > $O/$1.log 2>&1; }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM"
run c "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
python3 - <<PY
import csv, glob, collections
for tag in "abc":
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            if "pfb_" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            for c, v in d.items():
                print(f"{k:48s} {c:26s} {v / cnt[(k, c)]:.4g} per dispatch ({cnt[(k, c)]})")
PY
tail -2 $O/c.log | head -1
