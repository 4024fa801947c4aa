#!/bin/bash
# merged decode step of 8 sessions at prefix 96: per-kernel times, narrow dense layers (decode_wide_gemm=1) against the product
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for w in 1 0; do
rm -rf /tmp/prof_dec
export TAL_OPTIONS=decode_wide_gemm=$w GS=8
rocprofv3 --kernel-trace --stats -d /tmp/prof_dec -- python3 scripts/bench_greedy_step_multi.py 96 > /dev/null 2>&1
echo "== kernel stats, merged step of 8 sessions, prefix 96, decode_wide_gemm=$w"
python3 scripts/rocpd_summary.py $(find /tmp/prof_dec -name "*.db" | head -1) | head -24
done
