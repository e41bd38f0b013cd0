#!/bin/bash
# Dev tool (gpurun): LDS counters of the headline channelizer (cfg #3) + its step time
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_lds; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq_b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-others > $O/sq_b.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.path.expandvars('$GRAFT_REPO_ROOT/gpurun_out/r5_lds/sq_b/**/*counter_collection.csv'), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_LDS_IDX_ACTIVE': n[k] += 1
for k in acc:
    if 'pfb_spec' in k:
        print(k, 'launches', n[k], {c: '%.3e' % (v / max(1, n[k])) for c, v in acc[k].items()})
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R; for i in 1 2 3; do python3 bench.py --steps 20 --no-cpu --no-others 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg3 %.3f ms/step kernel %.3f ms fp32 %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['fp32_frac']))"; done
