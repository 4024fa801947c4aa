#!/bin/bash
# short clips end to end: new short-input kernels (product) against the previous dispatch (options), + kernel sequence of a 30 s clip
cd "$(dirname "$0")/.."
O=gpurun_out
{
echo "== product"; timeout 300 python scripts/bench_short.py 10 30 60 120 300 600
echo "== gemm_s64_below=0"; TAL_OPTIONS="gemm_s64_below=0" timeout 300 python scripts/bench_short.py 10 30 60 120 300 600
echo "== gconv_short_below=0"; TAL_OPTIONS="gconv_short_below=0" timeout 300 python scripts/bench_short.py 10 30 60 120 300 600
echo "== both off"; TAL_OPTIONS="gemm_s64_below=0,gconv_short_below=0" timeout 300 python scripts/bench_short.py 10 30 60 120 300 600
echo "== round-2 dispatch (all of round 3's short / medium-input choices off)"; TAL_OPTIONS="gemm_s64_below=0,gconv_short_below=0,gemm_no_n96=1,gemm_no_row_split=1" timeout 300 python scripts/bench_short.py 10 30 60 120 300 600
} > $O/r3_short_clips.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/short30 $R/$O/short300
REPS=3 timeout 300 rocprofv3 --kernel-trace -d $R/$O/short30 -- python3 $R/scripts/bench_short.py 30 > $R/$O/short30.log 2>&1
REPS=3 timeout 300 rocprofv3 --kernel-trace -d $R/$O/short300 -- python3 $R/scripts/bench_short.py 300 > $R/$O/short300.log 2>&1
cd $R
python scripts/rocpd_sequence.py $(find $O/short30 -name "*.db" | head -1) 52 > $O/r3_clip_30s_kernel_sequence.txt
python scripts/rocpd_sequence.py $(find $O/short300 -name "*.db" | head -1) 78 > $O/r3_clip_5min_kernel_sequence.txt
rm -rf $O/short30 $O/short300
cat $O/r3_short_clips.txt
