#!/bin/bash
# the rocprofv3 passes of scripts/profile_round.sh alone (kernel stats, FETCH / WRITE traffic, SQ and instruction counters of the
# default bench without its exact-fp32 pass), into gpurun_out/r4final/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode > $O/bench_1h_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_sq -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE -d $O/pmc_inst -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 > $O/pmc_inst.log 2>&1
cd $R
python scripts/pmc_sq_summary.py $(find $O/pmc_sq -name "*.db" | head -1) > $O/pmc_sq_all_kernels.txt
python scripts/pmc_generic.py $(find $O/pmc_inst -name "*.db" | head -1) tal > $O/pmc_inst_all_kernels.txt
S=$(find $O/stats -name "*.db" | head -1); F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python scripts/rocpd_summary.py $S > $O/bench_1h_kernel_stats.txt
python scripts/pmc_traffic_json.py $F $W > $O/pmc_traffic.json
(python scripts/rocpd_pmc.py $F tal::; python scripts/rocpd_pmc.py $W tal::) > $O/pmc_traffic_all_kernels.txt
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_inst
