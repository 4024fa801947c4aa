#!/bin/bash
# persistent tile loop (next tile's first operands requested before the epilogue) against one workgroup per tile
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_w64
mkdir -p $O
cd $R/scripts/ubench
for i in 1 2; do for b in w64_p0 w64_persist; do
  echo "=== $b" >> $O/ubench_persist.txt
  timeout 120 ./$b 2>&1 | grep -A1 "K loop\|max |err" >> $O/ubench_persist.txt
done; done
cat $O/ubench_persist.txt
