#!/bin/bash
# One GPU job: the 64 x 160-per-wave dense-layer K loop (scripts/ubench/gemm_f16x3_w64.hip) and its ablations beside the
# 32 x 160-per-wave kernel on the same box, then SQ counter passes of both (separate rocprofv3 runs, program after `--`).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_w64
mkdir -p $O
cd $R/scripts/ubench
for b in gemm_f16x3 w64_p0 w64_p1 w64_p2 w64_topbar w64_nvg6 w64_abl1 w64_abl8 w64_abl9 gemm_f16x3 w64_p0; do
  echo "=== $b" >> $O/ubench.txt
  timeout 120 ./$b >> $O/ubench.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for b in gemm_f16x3 w64_p0; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_sq_$b -- $R/scripts/ubench/$b > $O/pmc_sq_$b.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE -d $O/pmc_inst_$b -- $R/scripts/ubench/$b > $O/pmc_inst_$b.log 2>&1
  python3 $R/scripts/pmc_generic.py $(find $O/pmc_sq_$b -name "*.db" | head -1) gemm > $O/pmc_sq_$b.txt 2>&1
  python3 $R/scripts/pmc_generic.py $(find $O/pmc_inst_$b -name "*.db" | head -1) gemm > $O/pmc_inst_$b.txt 2>&1
  rm -rf $O/pmc_sq_$b $O/pmc_inst_$b
done
tail -n 200 $O/ubench.txt
