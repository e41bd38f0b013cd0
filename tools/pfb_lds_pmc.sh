#!/bin/bash
# LDS bank-conflict share of the channelizer kernels (run through gpurun).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/prof_pfb_lds -- python3 $R/tools/quick_wide.py btle40 zb16 > $O/prof_pfb_lds.log 2>&1
tail -n 2 $O/prof_pfb_lds.log | cut -c1-160
