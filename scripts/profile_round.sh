#!/bin/bash
# One GPU job that regenerates the round's measurement artefacts (run through gpurun; copy the results from
# gpurun_out/r1final/ into profiles/): bench lines (default + TAL_TDS_F32=1), rocprofv3 kernel stats, FETCH / WRITE
# traffic passes, SQ counter pass.  PMC passes are separate runs with --kernel-trace only, as the pool requires.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1final
mkdir -p $O
cd $R
python bench.py > $O/bench_1h.json 2> $O/bench_1h.err
TAL_TDS_F32=1 python bench.py --no-cpu-baseline > $O/bench_1h_fp32.json 2> $O/bench_1h_fp32.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_sq -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
cd $R
Q=$(find $O/pmc_sq -name "*.db" | head -1)
python scripts/pmc_sq_summary.py $Q > $O/pmc_sq_all_kernels.txt
S=$(find $O/stats -name "*.db" | head -1); F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python scripts/rocpd_summary.py $S > $O/kernel_stats.txt
python scripts/pmc_traffic_json.py $F $W > $O/pmc_traffic.json
(python scripts/rocpd_pmc.py $F tal::; python scripts/rocpd_pmc.py $W tal::) > $O/pmc_traffic_all_kernels.txt
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls -la $O
