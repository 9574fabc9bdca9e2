#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   1. kernel trace + stats of the default bench command  -> kernel durations
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes -> HBM traffic per launch
# Usage: bash tools/profile_gpu.sh <tag> ["extra bench.py args"]      (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
BENCH_ARGS=${2:-}      # e.g. "--shape large" for the configs[3] kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
echo "[profile] kernel trace"; 
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras $BENCH_ARGS > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log" || echo "trace run failed"
echo "[profile] pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$REPO/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-parity-gate --no-extras $BENCH_ARGS > "$OUT/bench_under_fetch.json" 2> "$OUT/fetch.log" || echo "fetch run failed"
echo "[profile] pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$REPO/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-parity-gate --no-extras $BENCH_ARGS > "$OUT/bench_under_write.json" 2> "$OUT/write.log" || echo "write run failed"
find "$OUT" -name "*.csv" | head -20
