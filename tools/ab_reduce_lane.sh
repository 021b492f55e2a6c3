#!/bin/bash
# same-box A/B of one compile-time switch of csrc/sr_bwd.hip on the SR training workloads: tools/ab_reduce_lane.sh "<flags of the variant>"
#   e.g. "-DWG_REDUCE_LANE=0" (reductions on the caller's stream), "-DWG_REDUCE_UNROLL=0" (one load per round trip in the reduction)
# product library and variant alternate, two rounds; outputs gpurun_out/ab_{sr,ref}_{product,variant}_{1,2}.log
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; V=$R/gpurun_out/variants; mkdir -p $V
NVSR_VARIANT_ONLY=sr_bwd.hip NVSR_EXTRA_HIPCC_FLAGS="$1" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/variant.so')" > $R/gpurun_out/ab_build.log 2>&1
cd $R
for round in 1 2; do
  for lib in product variant; do
    if [ $lib = variant ]; then export NVSR_HIP_LIB=$V/variant.so; else unset NVSR_HIP_LIB; fi
    timeout -k 10 200 python3 bench.py --workload sr --steps 4 --warmup 2 > gpurun_out/ab_sr_${lib}_$round.log 2>&1
    timeout -k 10 150 python3 bench.py --workload refine --refine-what sr --steps 10 --warmup 3 --no-split > gpurun_out/ab_ref_${lib}_$round.log 2>&1
    echo "$lib $round done"
  done
done
