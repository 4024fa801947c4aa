#!/bin/bash
# round-4 job 1: the default bench line with its new fields, the decode workload with its CPU leg, fresh FETCH / WRITE traffic
# passes (source-hashed), and the bounded RCCL probe against a communicator that really hangs (two ranks on this box's one GPU)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4job1
mkdir -p $O
cd $R
python bench.py > $O/bench_1h.json 2> $O/bench_1h.err
python bench.py --workload decode --steps 1 --warmup 1 > $O/bench_decode_1h_episode.json 2> $O/bench_decode.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-decode-episode --no-clip-latency --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
cd $R
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python scripts/pmc_traffic_json.py $F $W > $O/pmc_traffic.json
(python scripts/rocpd_pmc.py $F tal; python scripts/rocpd_pmc.py $W tal) > $O/pmc_traffic_all_kernels.txt
rm -rf $O/pmc_fetch $O/pmc_write
date +%s.%N > $O/probe_t0
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --segments 8 --no-one-gpu-reference > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err
echo "rc=$?" >> $O/bench_2ranks_one_gpu.err
date +%s.%N > $O/probe_t1
ls -la $O
