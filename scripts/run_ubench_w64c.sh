#!/bin/bash
# the two K loops under a duty cycle: a memset of n MB (a low-power phase) between the launches
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_w64
mkdir -p $O
cd $R/scripts/ubench
for mb in 0 1024 3072; do
  for b in gemm_f16x3 w64_p0; do
    echo "=== $b FILLER_MB=$mb" >> $O/ubench_duty.txt
    FILLER_MB=$mb timeout 120 ./$b 2>&1 | grep -A1 "K loop" >> $O/ubench_duty.txt
  done
done
cat $O/ubench_duty.txt
