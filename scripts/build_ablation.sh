#!/bin/bash
# Ablation builds of the library (kernel experiments only): build/abl/<name>.so with extra -D flags; select one with
# TAL_ASRD_LIB=build/abl/<name>.so.   usage: scripts/build_ablation.sh name -DFLAG [-DFLAG ...]
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
