#!/usr/bin/env bash
# Builds the kernels of a git revision (default HEAD) into tools/ab/libcogaim_<name>.so so that two builds can
# be timed on the SAME GPU box in one gpurun call (box-to-box spread is ~5-10%, larger than most kernel deltas):
#   tools/ab_build.sh head HEAD
#   gpurun -- 'for v in head ""; do COGAIM_LIB=${v:+tools/ab/libcogaim_$v.so} python tools/gemm_bench.py 2; done'
# The copy is built with -DCOGAIM_TOOLS (experiment knobs of cogaim_set_gemm_algo; the shipped library has none).
# COGAIM_LIB selects the library the ctypes binding loads (cognitive_aim_depth_estimation_amd/_lib.py).
set -euo pipefail
name=${1:?name}; rev=${2:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/pkg/csrc" "$tmp/include" "$root/tools/ab"
for f in $(git -C "$root" ls-tree --name-only "$rev" cognitive_aim_depth_estimation_amd/csrc/); do
  git -C "$root" show "$rev:$f" > "$tmp/pkg/csrc/$(basename "$f")"
done
git -C "$root" show "$rev:include/cogaim.h" > "$tmp/include/cogaim.h"      # small.hip includes ../../include/cogaim.h
objs=()
for src in "$tmp"/pkg/csrc/*.hip; do
  b=$(basename "$src" .hip); extra=""
  [ "$b" = attention ] && extra="-fno-honor-nans"
  timeout 900 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCOGAIM_TOOLS ${COGAIM_EXTRA_FLAGS:-} $extra -c "$src" -o "$tmp/$b.o" &
  objs+=("$tmp/$b.o")
done
wait
timeout 300 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab/libcogaim_$name.so" "${objs[@]}"
echo "built $root/tools/ab/libcogaim_$name.so from $rev"
