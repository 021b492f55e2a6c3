#!/bin/bash
# times every variant of scratch/variants/bl_*.so on the GPU box (tools/bwd_time_one.py: fine-pass backward alone, 4096 rays, S = 64 / 128), two alternations
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; V=$R/scratch/variants
for round in 1 2; do
  for so in ${LIBS:-$(ls $V/bl_*.so)}; do
    echo "round $round $(basename $so .so): $(NVSR_HIP_LIB=$so timeout -k 10 120 python3 $R/tools/bwd_time_one.py 2>&1 | tail -1)"
  done
done
