#!/bin/bash
# rocprofv3 kernel trace of the large shape's two-launch form (per-kernel durations of the three launches):
#   bash tools/trace_split.sh <tag> [kernel]     -> gpurun_out/trace_<tag>/
set -u
TAG=${1:-r05}
KERNEL=${2:-coop_split}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$REPO/bench.py" --shape large --kernel $KERNEL --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log" || echo "trace run failed"
find "$OUT" -name "*kernel_stats.csv" | head -3
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:100], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min_us", float(r["MinNs"]) / 1e3)
PY
