#!/bin/bash
# round 6, fourth GPU job: packed-FMA version of the fused first-stage conv, per-session fold flag, corpus decode
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "first_resize or sd_ or golden or mean_folded") > $O/pytest_parity.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_parity.txt
(timeout 1200 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_flow.py tests/test_gpu_episode.py -m gpu -x -q) > $O/pytest_decode.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_decode.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency > $O/bench_1h_fused.json 2> $O/bench_1h.err
TAL_OPTIONS=gconv_c1_fuse=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency > $O/bench_1h_unfused.json 2> /dev/null
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency > $O/bench_1h_fused_b.json 2> /dev/null
python scripts/bench_episode_streams.py 3600 8 2>&1 | grep -v amdgpu.ids > $O/episode_streams_8x1h.txt
grep -h "passed\|failed\|rc=" $O/pytest_parity.txt $O/pytest_decode.txt
tail -12 $O/episode_streams_8x1h.txt
