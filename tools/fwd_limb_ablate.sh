#!/bin/bash
# Where the time of the limb training forward goes (run on the GPU box): rebuilds with -DL3_ABLATE=<bits> (decode_limb.hip: 1 no plane
# gathers, 2 no gate words, 4 no wait for the weight copies, 8 no bias + ReLU) and prints the forward-only time of tools/train_step_time.py.
R=$GRAFT_REPO_ROOT
for bits in ${BITS:-0 1 2 4 8 15}; do
  NVSR_EXTRA_HIPCC_FLAGS="-DL3_ABLATE=$bits" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(force=True)" > /dev/null 2>&1
  echo "L3_ABLATE=$bits: $(python3 $R/tools/train_step_time.py planes 2>/dev/null | tail -1)"
done
python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(force=True)" > /dev/null 2>&1
