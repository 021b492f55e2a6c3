#!/bin/bash
# Where the time of the forward limb convolution goes (run on the GPU box).  Variant libraries are built beforehand with
#   NVSR_EXTRA_HIPCC_FLAGS="-DCV_ABLATE=<bits>" build_extension(out_path='scratch/variants/cv_<bits>.so')
# (sr.hip: 1 weight fragments of tap 0 only, 2 no patch loads, 4 no split + LDS writes -- wrong results by design).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for bits in ${BITS:-0 1 2 6 7}; do
  echo "CV_ABLATE=$bits: $(NVSR_HIP_LIB=$R/scratch/variants/cv_$bits.so python3 $R/tools/conv_time.py $SHAPE 2>/dev/null | grep bf16x3 | tr '\n' ' ')"
done
