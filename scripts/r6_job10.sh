#!/bin/bash
# what the driver does at round end, on the final tree: the -m gpu suite, smoke(), the bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6j
mkdir -p $O
cd $R
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
echo "smoke rc=$?" >> $O/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
grep -h "passed\|failed\|rc=" $O/pytest_gpu.txt $O/smoke.txt
grep smoke $O/smoke.txt
