#!/bin/bash
# soak: the contention / determinism tests five times in a row, the whole -m gpu suite twice -- a flake that shows once in 45 quiet runs
# (profiles/r5_head_lds_dma_race.txt) should show here
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6soak
mkdir -p $O
cd $R
for i in 1 2 3 4 5; do
  (timeout 900 python -m pytest tests/test_gpu_determinism.py -m gpu -q) > $O/determinism_$i.txt 2>&1
  echo "run $i rc=$?" >> $O/summary.txt
  tail -1 $O/determinism_$i.txt >> $O/summary.txt
done
for i in 1 2; do
  (timeout 1500 python -m pytest tests -m gpu -q) > $O/suite_$i.txt 2>&1
  echo "suite $i rc=$?" >> $O/summary.txt
  tail -1 $O/suite_$i.txt >> $O/summary.txt
done
cat $O/summary.txt
