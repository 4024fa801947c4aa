#!/bin/bash
# One GPU job that regenerates the round's measurement artefacts (run through gpurun; copy the results from
# gpurun_out/r6final/ into profiles/ with the r6_ prefix): bench lines of every workload, rocprofv3 kernel stats,
# FETCH / WRITE traffic passes, SQ counter passes, decode-step and short-clip traces.  PMC passes are separate runs with
# --kernel-trace only, as the pool requires; every profiled program is `python3 <script>` directly after `--`.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6final
mkdir -p $O
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_1h.json 2> $O/bench_1h.err     # (the driver's command)
TAL_OPTIONS=gemm_no_w64 python bench.py --no-cpu-baseline --no-exact-pass > $O/bench_1h_128row_tiles.json 2> /dev/null
TAL_OPTIONS=tds_exact_f32 python bench.py --no-cpu-baseline > $O/bench_1h_fp32.json 2> $O/bench_1h_fp32.err
TAL_OPTIONS=tds_fp32_activations python bench.py --no-cpu-baseline > $O/bench_1h_fp32_activations.json 2> /dev/null
python bench.py --workload segments --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_segments_64x5min.json 2> /dev/null
python bench.py --workload segments --segments 8 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_segments_8x5min.json 2> /dev/null
python bench.py --workload decode --steps 2 --warmup 1 > $O/bench_decode_1h_episode.json 2> /dev/null
python scripts/bench_gconv_grid.py > $O/gconv_grid.txt 2>&1
python scripts/bench_greedy_step_multi.py 32 2>&1 | grep -v amdgpu.ids > $O/decode_step_merged.txt
python scripts/bench_short.py 10 30 60 120 300 600 > $O/short_clips_product.txt 2>&1
python scripts/bench_greedy_step.py 1 16 32 48 64 96 128 256 > $O/decode_step.txt 2>&1
TAL_OPTIONS=decode_no_fold python scripts/bench_greedy_step.py 1 16 32 48 64 96 128 256 > $O/decode_step_unfolded_layer.txt 2>&1
TAL_OPTIONS=decode_no_fold python scripts/bench_greedy_step_multi.py 32 2>&1 | grep -v amdgpu.ids > $O/decode_step_merged_unfolded_layer.txt
python scripts/r6_uisrnn_predict.py 2>&1 | grep -v amdgpu.ids > $O/uisrnn_predict.txt
python scripts/bench_episode_streams.py 3600 8 2>&1 | grep -v amdgpu.ids > $O/episode_streams.txt
python scripts/bench_episode_streams.py 600 32 2>&1 | grep -v amdgpu.ids > $O/episode_streams_32x10min.txt
python scripts/bench_episode.py 300 2>&1 | grep -v amdgpu.ids > $O/episode_5min.txt
TAL_OPTIONS=decode_no_fold python scripts/bench_episode.py 300 2>&1 | grep -v amdgpu.ids > $O/episode_5min_unfolded_layer.txt
python scripts/bench_episode.py 3600 2>&1 | grep -v amdgpu.ids > $O/episode_1h.txt
cd /tmp && export TMPDIR=/tmp
# (only the 1-hour steps may be in the stats table: its averages are what roofline.avg_launch_ms is checked against)
rocprofv3 --kernel-trace --stats -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --no-clock-sampler > $O/bench_1h_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 --passes 1 --no-clock-sampler > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 --passes 1 --no-clock-sampler > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_sq -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 --passes 1 --no-clock-sampler > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE -d $O/pmc_inst -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 --passes 1 --no-clock-sampler > $O/pmc_inst.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/dstats -- python3 $R/scripts/bench_episode.py 300 > $O/decode_5min_under_rocprof.log 2>&1
REPS=3 rocprofv3 --kernel-trace -d $O/gstep -- python3 $R/scripts/bench_greedy_step.py 32 > $O/gstep.log 2>&1
rocprofv3 --kernel-trace -d $O/eptrace -- python3 $R/scripts/bench_episode.py 300 > $O/eptrace.log 2>&1
REPS=3 rocprofv3 --kernel-trace -d $O/short -- python3 $R/scripts/bench_short.py 30 > $O/short.log 2>&1
cd $R
Q=$(find $O/pmc_sq -name "*.db" | head -1); I=$(find $O/pmc_inst -name "*.db" | head -1)
python scripts/pmc_sq_summary.py $Q > $O/pmc_sq_all_kernels.txt
python scripts/pmc_generic.py $I tal > $O/pmc_inst_all_kernels.txt
S=$(find $O/stats -name "*.db" | head -1); F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python scripts/rocpd_summary.py $S > $O/bench_1h_kernel_stats.txt
if grep -q "gemm_s64_kernel" $O/bench_1h_kernel_stats.txt; then
    echo "profile_round.sh: the 1-hour kernel stats contain short-input dense launches (gemm_s64_kernel): the stats pass ran more than the 1-hour steps" >&2
    mv $O/bench_1h_kernel_stats.txt $O/bench_1h_kernel_stats.REJECTED.txt
    FAILED=1
fi
python scripts/pmc_traffic_json.py $F $W > $O/pmc_traffic.json
(python scripts/rocpd_pmc.py $F tal; python scripts/rocpd_pmc.py $W tal) > $O/pmc_traffic_all_kernels.txt
python scripts/rocpd_summary.py $(find $O/dstats -name "*.db" | head -1) > $O/decode_5min_kernel_stats.txt
python scripts/rocpd_sequence.py $(find $O/gstep -name "*.db" | head -1) 29 > $O/decode_step_U32_kernel_sequence.txt
python scripts/r6_episode_gaps.py $(find $O/eptrace -name "*.db" | head -1) > $O/episode_5min_step_breakdown.txt
python scripts/rocpd_sequence.py $(find $O/short -name "*.db" | head -1) 52 > $O/clip_30s_kernel_sequence.txt
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_inst $O/dstats $O/gstep $O/short $O/eptrace
ls -la $O
exit ${FAILED:-0}
