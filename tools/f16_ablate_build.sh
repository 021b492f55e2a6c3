#!/bin/bash
# builds -DR3_ABLATE=<bits> variants of render3.hip under scratch/variants/ (CPU container; the libraries travel to the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for bits in "$@"; do
  out=scratch/variants/r3_a$bits.so
  NVSR_VARIANT_ONLY="render3.hip" NVSR_EXTRA_HIPCC_FLAGS="-DR3_ABLATE=$bits" python -c "
import sys; sys.path.insert(0,'.')
import nvsr_amd
from nvsr_amd import build
build.build_extension(out_path='$out')
" || { echo "build failed for $bits"; rm -f $out; }
done
