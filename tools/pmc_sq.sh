#!/bin/bash
# SQ counter passes for the dominant kernel (issue / wait breakdown of a wave's cycles; MI355X_MICROARCH.md
# "rocprofv3 PMC slots": 8 SQ slots per pass, --pmc alone, no trace flags).
# Usage: bash tools/pmc_sq.sh <tag> ["extra bench.py args"]     (outputs under gpurun_out/pmc_<tag>/)
set -u
TAG=${1:-r04}
BENCH_ARGS=${2:-}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1 || true
i=0
for SET in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" ; do
  i=$((i+1))
  echo "[pmc] pass $i: $SET"
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/p$i" -o p$i -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity-gate --no-extras $BENCH_ARGS > "$OUT/p$i.json" 2> "$OUT/p$i.log" || echo "pass $i failed (see p$i.log)"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fccqp" not in k: continue
        short = "fused" if "fused_kernel" in k else ("coop" if "coop_kernel" in k else ("generic" if "generic" in k else k[:40]))
        res.setdefault(short, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summary = {k: {c: max(v) for c, v in d.items()} for k, d in res.items()}   # (the batch launch, not the empty fallback)
json.dump(summary, open(os.path.join(out, "sq_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
