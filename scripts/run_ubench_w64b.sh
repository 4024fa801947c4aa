#!/bin/bash
# does operand entropy set the K loop's time?  lo halves with 0 / 3 / 5 / 8 low mantissa bits cleared, and lo = 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_w64
mkdir -p $O
cd $R/scripts/ubench
for m in 0 3 5 8 16 0; do
  echo "=== w64_p0 lo_mask=$m" >> $O/ubench_lomask.txt
  timeout 120 ./w64_p0 $m >> $O/ubench_lomask.txt 2>&1
done
cat $O/ubench_lomask.txt
