#!/bin/bash
# Where the time of the limb backward goes (run on the GPU box): rebuilds the library with -DBL_ABLATE=<bits> (render_bwd_limb.hip: 1 no wait
# for the weight copies, 2 no gate masks, 4 no plane scatter / view rows, 8 no exposed limb splits) and times the planes-only train step.
R=$GRAFT_REPO_ROOT
for bits in ${BITS:-0 1 2 4 8 15}; do
  NVSR_EXTRA_HIPCC_FLAGS="-DBL_ABLATE=$bits" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(force=True)" > /dev/null 2>&1
  echo "BL_ABLATE=$bits: $(python3 $R/tools/train_step_time.py planes 2>/dev/null | head -1)"
done
python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(force=True)" > /dev/null 2>&1
