#!/bin/bash
# round 6, second GPU job: the decode-side tests on the folded layer, the GRU one-launch step, then the A/B numbers
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b
mkdir -p $O
cd $R
(time timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_flow.py tests/test_gpu_episode.py tests/test_uisrnn_host.py tests/test_gpu_distributed.py tests/test_gpu_half_audio.py tests/test_gpu_pool.py -m gpu -x -q --durations=10) > $O/pytest_decode.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_decode.txt
(timeout 600 python -m pytest tests/test_gpu_determinism.py -m gpu -x -q -k "decode or episode or decoder") > $O/pytest_determinism_decode.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_determinism_decode.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "b4 or sd_30s or mean_folded or head_on") > $O/pytest_parity_new.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_parity_new.txt
python scripts/bench_greedy_step.py 1 16 32 64 128 256 > $O/decode_step_folded.txt 2>&1
TAL_OPTIONS=decode_no_fold python scripts/bench_greedy_step.py 1 16 32 64 128 256 > $O/decode_step_unfolded.txt 2>&1
python scripts/bench_greedy_step_multi.py 32 2>&1 | grep -v amdgpu.ids > $O/decode_step_merged_folded.txt
TAL_OPTIONS=decode_no_fold python scripts/bench_greedy_step_multi.py 32 2>&1 | grep -v amdgpu.ids > $O/decode_step_merged_unfolded.txt
python scripts/bench_episode.py 300 > $O/episode_5min_folded.txt 2>&1
TAL_OPTIONS=decode_no_fold python scripts/bench_episode.py 300 > $O/episode_5min_unfolded.txt 2>&1
python scripts/r6_uisrnn_predict.py > $O/uisrnn_predict.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-pass --no-per-rank-share > $O/bench_1h.json 2> $O/bench_1h.err
tail -3 $O/pytest_decode.txt $O/pytest_determinism_decode.txt $O/pytest_parity_new.txt
cat $O/decode_step_folded.txt $O/decode_step_unfolded.txt
