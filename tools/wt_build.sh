#!/usr/bin/env bash
# Builds the WORKING TREE's kernels with -DCOGAIM_TOOLS (experiment knobs of cogaim_set_gemm_algo) into
# tools/ab/libcogaim_<name>.so, for same-box A/B timing next to the shipped library:
#   tools/wt_build.sh wt && gpurun -- 'COGAIM_LIB=tools/ab/libcogaim_wt.so python tools/gemm_bench.py 300 301'
set -euo pipefail
name=${1:?name}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
objs=()
for src in "$root"/cognitive_aim_depth_estimation_amd/csrc/*.hip; do
  b=$(basename "$src" .hip); extra=""
  [ "$b" = attention ] && extra="-fno-honor-nans"
  timeout 900 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCOGAIM_TOOLS ${COGAIM_EXTRA_FLAGS:-} $extra -c "$src" -o "$tmp/$b.o" &
  objs+=("$tmp/$b.o")
done
wait
mkdir -p "$root/tools/ab"
timeout 300 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab/libcogaim_$name.so" "${objs[@]}"
echo "built $root/tools/ab/libcogaim_$name.so"
