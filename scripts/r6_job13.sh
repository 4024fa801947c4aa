#!/bin/bash
# key chunks of the split cross-attention: ablation builds with 4 / 6 / 12 / 16 chunks against the product's 8
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6m
mkdir -p $O
cd $R
for n in 4 6 12 16; do bash scripts/build_ablation.sh nch$n -DTAL_SPLIT_NCH=$n > /dev/null 2>&1; done
for rep in 1 2; do
  echo "== product (8 chunks), rep $rep" >> $O/chunks.txt
  python scripts/bench_greedy_step.py 16 32 64 2>&1 | grep -v amdgpu.ids >> $O/chunks.txt
  python scripts/bench_episode.py 300 2>&1 | grep "rep 1" >> $O/chunks.txt
  for n in 4 6 12 16; do
    echo "== $n chunks, rep $rep" >> $O/chunks.txt
    TAL_ASRD_LIB=build/abl/nch$n.so python scripts/bench_greedy_step.py 16 32 64 2>&1 | grep -v amdgpu.ids >> $O/chunks.txt
    TAL_ASRD_LIB=build/abl/nch$n.so python scripts/bench_episode.py 300 2>&1 | grep "rep 1" >> $O/chunks.txt
  done
done
cat $O/chunks.txt
