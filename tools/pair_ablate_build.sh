#!/bin/bash
# builds timing variants of decode_pair.hip under scratch/variants/ (CPU container; the libraries travel to the GPU box):
#   tools/pair_ablate_build.sh name1 "-DDP_ABLATE=1" name2 "-DR3_ABLATE=32 -DDP_ABLATE=2" ...     -> scratch/variants/dp_<name>.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
FILES=${FILES:-decode_pair.hip}
while [ $# -ge 2 ]; do
  out=scratch/variants/dp_$1.so
  NVSR_VARIANT_ONLY="$FILES" NVSR_EXTRA_HIPCC_FLAGS="$2" python -c "
import sys; sys.path.insert(0,'.')
import nvsr_amd
from nvsr_amd import build
build.build_extension(out_path='$out')
" > /dev/null || { echo "build failed for $1"; rm -f $out; }
  shift 2
done
ls -la scratch/variants/dp_*.so
