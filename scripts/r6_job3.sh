#!/bin/bash
# round 6, third GPU job: the fused first-stage conv (parity + timing), the folded decoder layer with its row limit
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -x -q) > $O/pytest_parity.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_parity.txt
(timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_flow.py tests/test_gpu_episode.py -m gpu -x -q) > $O/pytest_decode.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_decode.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-pass --no-per-rank-share > $O/bench_1h_fused.json 2> $O/bench_1h.err
TAL_OPTIONS=gconv_c1_fuse=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode > $O/bench_1h_unfused.json 2> /dev/null
python scripts/bench_greedy_step.py 1 16 32 48 64 96 128 256 > $O/decode_step_folded.txt 2>&1
TAL_OPTIONS=decode_fold_rows=512 python scripts/bench_greedy_step.py 1 16 32 48 64 96 128 256 > $O/decode_step_folded_always.txt 2>&1
TAL_OPTIONS=decode_no_fold python scripts/bench_greedy_step.py 1 16 32 48 64 96 128 256 > $O/decode_step_unfolded.txt 2>&1
python scripts/bench_greedy_step_multi.py 32 2>&1 | grep -v amdgpu.ids > $O/decode_step_merged_folded.txt
TAL_OPTIONS=decode_no_fold python scripts/bench_greedy_step_multi.py 32 2>&1 | grep -v amdgpu.ids > $O/decode_step_merged_unfolded.txt
python scripts/bench_episode.py 300 > $O/episode_5min_folded.txt 2>&1
python scripts/bench_short.py 30 300 > $O/short_clips.txt 2>&1
grep -h "passed\|failed\|rc=" $O/pytest_parity.txt $O/pytest_decode.txt
paste $O/decode_step_folded.txt $O/decode_step_folded_always.txt $O/decode_step_unfolded.txt
