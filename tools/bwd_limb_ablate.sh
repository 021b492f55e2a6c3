#!/bin/bash
# Where the time of the limb backward goes (run on the GPU box): builds VARIANT libraries with -DBL_ABLATE=<bits> (render_bwd_limb.hip: 1 no
# wait for the weight copies, 2 no gate masks, 4 no plane scatter / view rows, 8 no exposed limb splits -- wrong results by design, so they
# never replace the product library: build_extension(out_path=...) + NVSR_HIP_LIB) and times the planes-only train step.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/gpurun_out/variants; mkdir -p $V
for bits in ${BITS:-0 1 2 4 8 15}; do
  NVSR_EXTRA_HIPCC_FLAGS="-DBL_ABLATE=$bits" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/bl_$bits.so')" > /dev/null 2>&1
  echo "BL_ABLATE=$bits: $(NVSR_HIP_LIB=$V/bl_$bits.so python3 $R/tools/train_step_time.py planes 2>/dev/null | head -1)"
done
