#!/bin/bash
# round 5, after the split head: A / B of the head input form, kernel sequences of a 30-second / 5-minute clip, short-clip latencies
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
python scripts/r5_ab_split_head.py > $O/split_head_ab.txt 2>&1; cat $O/split_head_ab.txt
python scripts/r5_head_embed_f16x3.py > $O/head_embed_f16x3.txt 2>&1; cat $O/head_embed_f16x3.txt
cd /tmp; export TMPDIR=/tmp
for S in 30 300; do
  REPS=3 rocprofv3 --kernel-trace -d $O/short$S -- python3 $R/scripts/bench_short.py $S > $O/short$S.log 2>&1
  python3 $R/scripts/rocpd_sequence.py $(find $O/short$S -name "*.db" | head -1) 52 > $O/clip_${S}s_kernel_sequence.txt
  rm -rf $O/short$S
done
cd $R; python scripts/bench_short.py 10 30 60 120 300 600 > $O/short_clips.txt 2>&1
cat $O/short_clips.txt
