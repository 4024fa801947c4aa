#!/bin/bash
# same-box A/B of the corpus decode: folded decoder layer (default) against decode_no_fold
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6h
mkdir -p $O
cd $R
for rep in 1 2; do
MODES="4x1,4x2,2x4,4x4,default" python scripts/bench_episode_streams.py 3600 8 2>&1 | grep -v amdgpu.ids > $O/streams_8x1h_folded_$rep.txt
MODES="4x1,4x2,2x4,4x4,default" TAL_OPTIONS=decode_no_fold python scripts/bench_episode_streams.py 3600 8 2>&1 | grep -v amdgpu.ids > $O/streams_8x1h_unfolded_$rep.txt
done
MODES="2x8,4x4,2x16,default" python scripts/bench_episode_streams.py 600 32 2>&1 | grep -v amdgpu.ids > $O/streams_32x10min_folded.txt
MODES="2x8,4x4,2x16,default" TAL_OPTIONS=decode_no_fold python scripts/bench_episode_streams.py 600 32 2>&1 | grep -v amdgpu.ids > $O/streams_32x10min_unfolded.txt
tail -n +1 $O/*.txt
