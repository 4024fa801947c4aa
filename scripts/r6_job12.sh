#!/bin/bash
# packed rows in merged decode steps: bit-identity tests, then A/B against option decode_no_pack
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6l
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_episode.py tests/test_gpu_flow.py tests/test_gpu_decoder.py -m gpu -x -q) > $O/pytest_decode.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_decode.txt
(timeout 600 python -m pytest tests/test_gpu_determinism.py -m gpu -x -q -k "decode or episode") > $O/pytest_determinism.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_determinism.txt
python scripts/bench_greedy_step_multi.py 32 2>&1 | grep -v amdgpu.ids > $O/merged_packed.txt
TAL_OPTIONS=decode_no_pack python scripts/bench_greedy_step_multi.py 32 2>&1 | grep -v amdgpu.ids > $O/merged_padded.txt
python scripts/bench_greedy_step_multi.py 20 2>&1 | grep -v amdgpu.ids > $O/merged20_packed.txt
TAL_OPTIONS=decode_no_pack python scripts/bench_greedy_step_multi.py 20 2>&1 | grep -v amdgpu.ids > $O/merged20_padded.txt
MODES="4x2,2x4,4x4,2x8,default" python scripts/bench_episode_streams.py 3600 8 2>&1 | grep -v amdgpu.ids > $O/streams_8x1h_packed.txt
MODES="4x2,2x4,4x4,2x8,default" TAL_OPTIONS=decode_no_pack python scripts/bench_episode_streams.py 3600 8 2>&1 | grep -v amdgpu.ids > $O/streams_8x1h_padded.txt
MODES="4x4,2x8,2x16,default" python scripts/bench_episode_streams.py 600 32 2>&1 | grep -v amdgpu.ids > $O/streams_32x10min_packed.txt
MODES="4x4,2x8,2x16,default" TAL_OPTIONS=decode_no_pack python scripts/bench_episode_streams.py 600 32 2>&1 | grep -v amdgpu.ids > $O/streams_32x10min_padded.txt
grep -h "passed\|failed\|rc=" $O/pytest_decode.txt $O/pytest_determinism.txt
tail -n +1 $O/merged*.txt $O/streams*.txt
