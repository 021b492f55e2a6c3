#!/bin/bash
# kernel traces of the SR stage with conv3x3_limb16_kernel forced to 2 / 3 / 4 rows per tile and with the launcher's cost model (see sr_rows_per_layer.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for t in auto 2 3 4; do
  if [ $t = auto ]; then unset NVSR_CV16_ROWS; else export NVSR_CV16_ROWS=$t; fi
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/srrows_$t -- python3 $R/bench.py --workload sr --steps 3 --warmup 1 --no-cpu-baseline --no-modes > /dev/null 2>&1
done
python3 $R/tools/sr_rows_per_layer.py
