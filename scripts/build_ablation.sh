#!/bin/bash
# Ablation builds of the library (kernel experiments only): build/abl/<name>.so with extra -D flags (layout constants such as
# -DGC_P10=10, -DGC_P18=0, -DGC_LB18=3); select one with TAL_ASRD_LIB=build/abl/<name>.so.
# The wrong-result ablation blocks of round 2 (GEMM_ABL_NOEPI / _STORESMALL, GC_ABL_NOSTORE / _ONE_STORE / _NOTAIL: what the
# epilogues and store phases cost, profiles/r2_gemm_f16x3_fit.txt, r2_gconv_ablations.txt) are no longer in the product
# sources; check out commit ab04c04 to rebuild them.
# usage: scripts/build_ablation.sh name -DFLAG [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/abl/$name
objs=""
for s in tal_asrd_amd/csrc/*.hip; do
  o=build/abl/$name/$(basename $s).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $s -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/$name.so $objs
echo built build/abl/$name.so
