#!/bin/bash
# Where the time of the limb training forward goes (run on the GPU box): builds VARIANT libraries with -DL3_ABLATE=<bits> (decode_limb.hip: 1
# no plane gathers, 2 no gate words, 4 no wait for the weight copies, 8 no bias + ReLU; separate files loaded through NVSR_HIP_LIB, the
# product library is never touched) and prints the forward-only time of tools/train_step_time.py.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/gpurun_out/variants; mkdir -p $V
for bits in ${BITS:-0 1 2 4 8 15}; do
  NVSR_EXTRA_HIPCC_FLAGS="-DL3_ABLATE=$bits" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/l3_$bits.so')" > /dev/null 2>&1
  echo "L3_ABLATE=$bits: $(NVSR_HIP_LIB=$V/l3_$bits.so python3 $R/tools/train_step_time.py planes 2>/dev/null | tail -1)"
done
