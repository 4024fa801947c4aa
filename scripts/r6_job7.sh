#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_flow.py tests/test_gpu_episode.py tests/test_gpu_decoder.py tests/test_gpu_half_audio.py -m gpu -x -q) > $O/pytest_decode.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_decode.txt
(timeout 600 python -m pytest tests/test_gpu_determinism.py -m gpu -x -q -k "decode or episode or decoder") > $O/pytest_determinism_decode.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_determinism_decode.txt
python scripts/bench_episode.py 300 > $O/episode_5min.txt 2>&1
python scripts/bench_episode.py 3600 > $O/episode_1h.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-clip-latency > $O/bench_1h.json 2> /dev/null
grep -h "passed\|failed\|rc=" $O/pytest_decode.txt $O/pytest_determinism_decode.txt
grep -h "rep 1" $O/episode_5min.txt $O/episode_1h.txt
