#!/bin/bash
# SQ counters of the short-input kernels on a 30-second clip (separate --pmc passes, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/sp1 $O/sp2
REPS=5 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/sp1 -- python3 $R/scripts/bench_short.py 30 > $O/sp1.log 2>&1
REPS=5 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE -d $O/sp2 -- python3 $R/scripts/bench_short.py 30 > $O/sp2.log 2>&1
cd $R
{ python scripts/pmc_sq_summary.py $(find $O/sp1 -name "*.db" | head -1); echo; python scripts/pmc_generic.py $(find $O/sp2 -name "*.db" | head -1) tal; } > $O/r3_pmc_clip_30s_all_kernels.txt 2>&1
rm -rf $O/sp1 $O/sp2
grep -i "s64\|kernel \|gconv_mfma" $O/r3_pmc_clip_30s_all_kernels.txt | cut -c1-200
