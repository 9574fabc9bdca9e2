#!/bin/bash
# One GPU-box session while developing: quick kernel timings (both kernels, phase stamps), then the
# whole GPU test suite.  Outputs under gpurun_out/.   Usage: bash tools/gpu_check.sh <tag>
set -u
TAG=${1:-dev}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 300 python tools/quick_bench.py 4096 W-sat > "$OUT/quick_sat.log" 2>&1 || echo "quick W-sat failed"
tail -4 "$OUT/quick_sat.log"
timeout -k 10 300 python tools/quick_bench.py 4096 W-psd > "$OUT/quick_psd.log" 2>&1 || echo "quick W-psd failed"
tail -4 "$OUT/quick_psd.log"
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > "$OUT/pytest.log" 2>&1
echo "pytest rc $?"
tail -15 "$OUT/pytest.log"
