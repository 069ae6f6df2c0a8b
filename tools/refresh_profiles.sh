#!/bin/bash
# Regenerates the profiles/ artefacts that belong to the current kernel sources.
#   on the GPU box (repo root):   bash tools/refresh_profiles.sh gpu gpurun_out/refresh
#   afterwards, where the results were merged back:  bash tools/refresh_profiles.sh collect gpurun_out/refresh r2
set -eu
MODE=$1; OUT=$2
if [ "$MODE" = gpu ]; then
  mkdir -p "$OUT"
  bash tools/profile_bench.sh "$OUT/prof"
  bash tools/pmc.sh "$OUT/pmc"
  bash tools/pmc_stack.sh "$OUT/pmc_stack"
  timeout -k 10 300 python3 bench.py --config D > "$OUT/benchD.json" 2> "$OUT/benchD.err"
  timeout -k 10 300 python3 bench.py --config D --two-pass-sums > "$OUT/benchD_two_pass.json" 2> "$OUT/benchD_two_pass.err"
  timeout -k 10 400 python3 tools/bench_configs.py > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
  echo "gpu part done: run the collect step, then bench.py once more for the line with measured traffic"
else
  TAG=$3
  python3 tools/hbm_traffic.py "$OUT/prof" profiles/hbm_traffic.json > /dev/null
  cp profiles/hbm_traffic.json profiles/${TAG}_hbm_traffic.json
  cp "$OUT"/prof/trace/*/*_kernel_stats.csv profiles/${TAG}_final_kernel_stats.csv
  python3 tools/pmc_summary.py "$OUT/pmc" "fused_kernel<0, 13, short" > profiles/${TAG}_pmc_summary.txt
  python3 tools/pmc_summary.py "$OUT/pmc_stack" "fused_kernel<6" > profiles/${TAG}_configC_hbm_counters.txt
  cp "$OUT/benchD.json" profiles/${TAG}_configD_bench.json
  cp "$OUT/benchD_two_pass.json" profiles/${TAG}_configD_two_pass_bench.json
  cp "$OUT/configs.jsonl" profiles/${TAG}_configs.jsonl
  echo "collected into profiles/"
fi
