#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (scripts/ubench/fetch_calib.hip), separate --pmc passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $O/fc_$C -- $R/scripts/ubench/fetch_calib > $O/fc_$C.log 2>&1
  python3 $R/scripts/rocpd_pmc.py $(find $O/fc_$C -name "*.db" | head -1) "_kernel\|read_rows\|write_b128" > /dev/null 2>&1
  python3 $R/scripts/rocpd_pmc.py $(find $O/fc_$C -name "*.db" | head -1) "" > $O/fetch_calib_$C.txt 2>&1
  rm -rf $O/fc_$C
done
cat $O/fc_FETCH_SIZE.log | tail -2; cat $O/fetch_calib_FETCH_SIZE.txt $O/fetch_calib_WRITE_SIZE.txt
