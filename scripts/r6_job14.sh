cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6n
for rep in 1 2; do for r in 32 48 64 96 128; do echo "fold_rows=$r: $(TAL_OPTIONS=decode_fold_rows=$r python scripts/bench_episode.py 3600 2>&1 | grep 'rep 1' | sed 's/ | SD pass.*//')" >> gpurun_out/r6n/fold_rows.txt; done; done
cat gpurun_out/r6n/fold_rows.txt
