#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench command (HIP graphs off so that kernel names stay per launch)
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the same workload, --kernel-trace only (never with sys/hip traces)
# Outputs land under gpurun_out/<tag>_{stats,fetch,write}; tools/summarize_profile.py turns them into profiles/<tag>_*.
# Usage: bash tools/profile_round.sh r01b
set -eo pipefail
TAG=${1:-r01b}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-graph"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -o runc -- $BENCH > "$OUT/${TAG}_stats.log" 2>&1
echo "stats pass done"
BENCH1="python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra --no-graph"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_fetch" -o runc -- $BENCH1 > "$OUT/${TAG}_fetch.log" 2>&1
echo "fetch pass done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_write" -o runc -- $BENCH1 > "$OUT/${TAG}_write.log" 2>&1
echo "write pass done"
grep -h '^{"metric' "$OUT/${TAG}_stats.log" > "$OUT/${TAG}_bench.json" || true
