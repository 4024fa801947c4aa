#!/bin/bash
# round 6, first GPU job: the whole -m gpu suite on the new tree, the default bench line (median of 3 passes + clock sampler),
# the same with 7 passes of 20 steps (where does a slow first pass come from?), what amdsmi reports on this box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a
mkdir -p $O
cd $R
python - > $O/amdsmi_probe.txt 2>&1 <<'PY'
import amdsmi, json
amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()
print("handles", len(h))
m = amdsmi.amdsmi_get_gpu_metrics_info(h[0])
print({k: v for k, v in m.items() if "clk" in k or "power" in k or "temp" in k or "throttle" in k or "activity" in k})
PY
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=25) > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 3 > $O/bench_1h.json 2> $O/bench_1h.err
python bench.py --steps 20 --warmup 3 --passes 7 --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-clip-latency > $O/bench_1h_7passes.json 2> /dev/null
python bench.py --steps 20 --warmup 3 --passes 7 --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-clip-latency --no-decode-episode > $O/bench_1h_7passes_no_decode_episode.json 2> /dev/null
python bench.py --steps 20 --warmup 3 --passes 7 --no-clock-sampler --no-cpu-baseline --no-exact-pass --no-per-rank-share --no-clip-latency --no-decode-episode > $O/bench_1h_7passes_no_sampler.json 2> /dev/null
tail -5 $O/pytest_gpu.txt
