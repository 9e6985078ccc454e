#!/usr/bin/env bash
# Runs ON THE GPU BOX: the harness and config-5 figures kept under profiles/ (tools/summarize... not needed: JSON lines).
#   gpurun -- 'bash tools/round_measurements.sh r3'
set -uo pipefail
tag=${1:?tag}
out=gpurun_out/${tag}_meas
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 200 python3 tools/harness_bench.py --n 1024 --src 518 > "$out/harness_518.json" 2> "$out/harness_518.err" || { echo "harness 518 failed"; tail -3 "$out/harness_518.err"; exit 1; }
timeout -k 10 200 python3 tools/harness_bench.py --n 96 --src 4032x3024 > "$out/harness_12mp.json" 2> "$out/harness_12mp.err" || { echo "harness 12mp failed"; tail -3 "$out/harness_12mp.err"; exit 1; }
for b in 1 4 16; do
  timeout -k 10 240 python3 bench.py --size 1036 --iters 8 --dtype fp16 --batch $b --steps 10 --warmup 3 --no-second-dtype --no-pipelined --no-cpu-baseline > "$out/config5_b$b.json" 2> "$out/config5_b$b.err" || { echo "config5 b$b failed"; tail -3 "$out/config5_b$b.err"; exit 1; }
done
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$out/config5_trace" -o t --output-format csv -- python3 bench.py --size 1036 --iters 8 --dtype fp16 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --no-second-dtype --no-kernel-profile --no-pipelined > "$out/config5_trace.log" 2>&1 || { echo "config5 trace failed"; tail -3 "$out/config5_trace.log"; exit 1; }
timeout -k 10 120 python3 tools/sweep_bench.py > "$out/sweep.json" 2> "$out/sweep.err" || { echo "sweep failed"; tail -3 "$out/sweep.err"; exit 1; }
echo done
