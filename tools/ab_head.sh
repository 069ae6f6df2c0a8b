#!/bin/bash
# Build the library of the last commit next to the working tree's, for same-run A/B timing:
#   bash tools/ab_head.sh && gpurun -- 'python tools/run_forward.py --lib mfcc_amd/lib/libmfcc_amd.so --lib build/variants/libmfcc_head.so'
set -e
rm -rf /tmp/mfcc_head_wt
git worktree add -q /tmp/mfcc_head_wt HEAD
(cd /tmp/mfcc_head_wt && make all >/dev/null 2>&1)
mkdir -p build/variants
cp /tmp/mfcc_head_wt/mfcc_amd/lib/libmfcc_amd.so build/variants/libmfcc_head.so
git worktree remove --force /tmp/mfcc_head_wt
echo "built build/variants/libmfcc_head.so from $(git rev-parse --short HEAD)"
