#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
REPS=3 rocprofv3 --kernel-trace -d $O/gstep -- python3 $R/scripts/bench_greedy_step.py 32 > $O/gstep.log 2>&1
rocprofv3 --kernel-trace -d $O/ep -- python3 $R/scripts/bench_episode.py 300 > $O/ep.log 2>&1
cd $R
python scripts/rocpd_sequence.py $(find $O/gstep -name "*.db" | head -1) 29 > $O/decode_step_U32_kernel_sequence.txt
python scripts/r6_episode_gaps.py $(find $O/ep -name "*.db" | head -1) > $O/episode_5min_step_breakdown.txt
rm -rf $O/gstep $O/ep
cat $O/decode_step_U32_kernel_sequence.txt $O/episode_5min_step_breakdown.txt
