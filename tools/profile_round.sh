#!/usr/bin/env bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh <tag>'): one bench line and the rocprofv3 passes the
# files under profiles/ are made from, all from the same build on the same box.  Each pass is its own run
# (kernel trace + stats; FETCH_SIZE; WRITE_SIZE; matrix-pipe counters), each under its own time limit.
# Afterwards, in the build container:  python tools/summarize_profile.py <tag> gpurun_out/<tag>
set -uo pipefail
tag=${1:?tag}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp 2>/dev/null && cd - >/dev/null
export TMPDIR=/tmp
B="--steps 3 --warmup 1 --no-cpu-baseline --no-second-dtype --no-kernel-profile --no-pipelined $*"
timeout -k 10 300 python3 bench.py $* > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$out/trace" -o t --output-format csv -- python3 bench.py $B > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o t --output-format csv -- python3 bench.py $B > "$out/fetch.log" 2>&1 || { echo "fetch failed"; tail -5 "$out/fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o t --output-format csv -- python3 bench.py $B > "$out/write.log" 2>&1 || { echo "write failed"; tail -5 "$out/write.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$out/mfma" -o t --output-format csv -- python3 bench.py $B > "$out/mfma.log" 2>&1 || { echo "mfma pass failed"; tail -5 "$out/mfma.log"; exit 1; }
find "$out" -name "*.csv" | head -20
echo done
