#!/bin/bash
# LDS counters of the grouped-conv kernels: product (K permutation) against build/abl/gc_noperm.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/gcp1 $O/gcp2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $O/gcp1 -- python3 $R/scripts/bench_gconv_split.py > $O/gcp1.log 2>&1
export TAL_ASRD_LIB=$R/build/abl/gc_noperm.so
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $O/gcp2 -- python3 $R/scripts/bench_gconv_split.py > $O/gcp2.log 2>&1
cd $R
{ echo "== K permutation (product)"; python scripts/pmc_generic.py $(find $O/gcp1 -name "*.db" | head -1) gconv_mfma; echo "== plain K order (build/abl/gc_noperm.so)"; python scripts/pmc_generic.py $(find $O/gcp2 -name "*.db" | head -1) gconv_mfma; } > $O/r3_gconv_kperm_lds_counters.txt 2>&1
rm -rf $O/gcp1 $O/gcp2
cut -c1-66,75-260 $O/r3_gconv_kperm_lds_counters.txt | grep "ELi2ELb1EE\|kernel\|=="
