#!/bin/bash
# round 6, fifth GPU job: the whole -m gpu suite on the tree with the solo decode loop inside the library, then decode numbers
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6e
mkdir -p $O
cd $R
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15) > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.txt
python scripts/bench_episode.py 300 > $O/episode_5min.txt 2>&1
TAL_OPTIONS=decode_no_fold python scripts/bench_episode.py 300 > $O/episode_5min_unfolded.txt 2>&1
python scripts/bench_episode.py 3600 > $O/episode_1h.txt 2>&1
python bench.py --steps 20 --warmup 3 > $O/bench_1h.json 2> $O/bench_1h.err
grep -h "passed\|failed\|rc=" $O/pytest_gpu.txt
grep -h "rep 1" $O/episode_5min.txt $O/episode_5min_unfolded.txt $O/episode_1h.txt
