#!/bin/bash
# Round artefacts for profiles/: kernel-trace stats of bench.py and the two HBM-traffic
# counter passes (each pass is its own rocprofv3 run; counters are never combined with traces).
# Run on the GPU box from the repo root:  bash tools/profile_bench.sh gpurun_out/prof_TAG
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/trace" -- python3 "$REPO/bench.py" --no-cpu-baseline > "$REPO/$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$REPO/$OUT/trace.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$REPO/$OUT/fetch" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$REPO/$OUT/fetch.log" 2>&1 || { echo "fetch failed"; tail -5 "$REPO/$OUT/fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$REPO/$OUT/write" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$REPO/$OUT/write.log" 2>&1 || { echo "write failed"; tail -5 "$REPO/$OUT/write.log"; exit 1; }
echo "profile done"
