#!/bin/bash
# round-4 job 2: where an 8-segment batched call (one rank's share of configs[3] at 8 GPUs) loses against the 64-segment call
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4job2
mkdir -p $O
cd $R
for n in 8 64; do
python bench.py --workload segments --segments $n --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_segments_$n.json 2> /dev/null
done
cd /tmp && export TMPDIR=/tmp
for n in 8 64; do
rocprofv3 --kernel-trace --stats -d $O/stats$n -- python3 $R/bench.py --workload segments --segments $n --no-cpu-baseline --no-prof --steps 5 --warmup 2 > $O/rocprof_$n.log 2>&1
python $R/scripts/rocpd_summary.py $(find $O/stats$n -name "*.db" | head -1) > $O/segments_${n}_kernel_stats.txt
rm -rf $O/stats$n
done
ls -la $O
