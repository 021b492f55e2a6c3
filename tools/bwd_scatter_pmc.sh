#!/bin/bash
# WRITE_SIZE / atomic-request A/B of the plane scatter of render_pass_backward_gates_limb_kernel (run on the GPU box): builds variant libraries
# with -DBL_SCATTER=0 (one set of 4 atomics per point), 1 (per run of points in one cell: rounds 1-2), 2 (every texel of a tile once: the
# product) and runs `bench.py --workload train` under rocprofv3 --pmc for each.  Un-merged payload of the fine pass = 4096 rays x 128 samples
# x 3 planes x 4 taps x 192 B = 1.208 GB; WRITE_SIZE is in KB.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/gpurun_out/variants; mkdir -p $V
export PMC_SCRIPT=../bench.py PMC_ARGS="--workload train --steps 3 --warmup 1 --no-cpu-baseline"
for m in ${MODES:-0 1 2}; do
  rm -f $V/bl_scatter_$m.so
  NVSR_EXTRA_HIPCC_FLAGS="-DBL_SCATTER=$m" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/bl_scatter_$m.so')" > $V/bl_scatter_$m.log 2>&1 || { echo "BL_SCATTER=$m: build failed"; continue; }
  export NVSR_HIP_LIB=$V/bl_scatter_$m.so
  bash $R/tools/pmc.sh sc${m}w WRITE_SIZE TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum || exit 1
  bash $R/tools/pmc.sh sc${m}f FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
  echo "== BL_SCATTER=$m"
  python3 $R/tools/pmc_read.py --kernel render_pass_backward_gates_limb sc${m}w sc${m}f
  unset NVSR_HIP_LIB
done
