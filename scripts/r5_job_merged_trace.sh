#!/bin/bash
# per-kernel durations of a decode step: one session / eight sessions in shared launches (uniform and mixed prefixes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for M in solo merged8 mixed8; do
  MODE=$M rocprofv3 --kernel-trace -d $O/mt_$M -- python3 $R/scripts/r5_merged_step_trace.py > $O/mt_$M.log 2>&1
  python3 $R/scripts/rocpd_summary.py $(find $O/mt_$M -name "*.db" | head -1) > $O/merged_trace_$M.txt
  rm -rf $O/mt_$M
done
head -30 $O/merged_trace_solo.txt $O/merged_trace_merged8.txt $O/merged_trace_mixed8.txt | cut -c1-150
