#!/bin/bash
# The gate-driven limb backward ALONE (tools/bwd_ablate.py: 4096 rays x 128 samples, planes 200^2) with -DBL_ABLATE=<bits> variant libraries:
# 1 no wait for the weight copies, 2 no gate masks, 4 no plane scatter / view rows, 8 no exposed limb splits, 16 scatter loop without its
# atomics.  Wrong results by design; never the product library.     BITS="0 4 16" bash tools/bwd_limb_kernel_ablate.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/gpurun_out/variants; mkdir -p $V
for bits in ${BITS:-0 1 2 4 8 16 15}; do
  NVSR_EXTRA_HIPCC_FLAGS="-DBL_ABLATE=$bits $EXTRA" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/blk_$bits.so')" > $V/blk_$bits.log 2>&1 || { echo "BL_ABLATE=$bits: build failed"; continue; }
  echo "BL_ABLATE=$bits $EXTRA: $(NVSR_HIP_LIB=$V/blk_$bits.so python3 $R/tools/bwd_ablate.py 2>/dev/null | grep -E 'no scatter|all, row ws' | tr '\n' ' ')"
done
