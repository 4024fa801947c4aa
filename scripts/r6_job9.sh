#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6i
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_episode.py tests/test_gpu_flow.py -m gpu -x -q) > $O/pytest_episode.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_episode.txt
python scripts/bench_episode_streams.py 3600 8 2>&1 | grep -v amdgpu.ids > $O/episode_streams.txt
python scripts/bench_episode_streams.py 600 32 2>&1 | grep -v amdgpu.ids > $O/episode_streams_32x10min.txt
grep -h "passed\|failed\|rc=" $O/pytest_episode.txt
tail -4 $O/episode_streams.txt $O/episode_streams_32x10min.txt
