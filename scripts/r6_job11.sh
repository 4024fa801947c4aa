#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6k
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -x -q) > $O/pytest_parity.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_parity.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode > $O/bench_1h.json 2> /dev/null
TAL_OPTIONS=gconv_long_tt=256 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode > $O/bench_1h_tt256.json 2> /dev/null
python bench.py --workload segments --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_segments.json 2> /dev/null
TAL_OPTIONS=gconv_long_tt=256 python bench.py --workload segments --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_segments_tt256.json 2> /dev/null
grep -h "passed\|failed\|rc=" $O/pytest_parity.txt
