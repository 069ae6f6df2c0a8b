#!/bin/bash
# Counter passes for the one-launch stacked-spectrum path (config C shape); each pass its own run.
set -u
OUT=${1:-gpurun_out/pmc_stack}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$REPO/$OUT/$name" -- python3 "$REPO/tools/stackc_bench.py" > "$REPO/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$REPO/$OUT/$name.log"; return 1; }
}
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE && \
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum && \
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
echo "pmc done"
