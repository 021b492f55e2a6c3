#!/bin/bash
# Where the time of the SR weight-gradient contraction goes (run on the GPU box).  Variant libraries are built beforehand with
#   NVSR_EXTRA_HIPCC_FLAGS="-DWG_TUNE -DWG_ABLATE=<bits>" build_extension(out_path='scratch/variants/wg_<bits>.so')
# (sr_bwd.hip: 1 no global fetches, 2 no split + LDS writes, 4 no MFMAs, 8 no fragment reads -- wrong results by design).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for bits in ${BITS:-0 1 2 3 4 8}; do
  echo "WG_ABLATE=$bits: $(NVSR_HIP_LIB=$R/scratch/variants/wg_$bits.so python3 $R/tools/conv_wgrad_time.py 2>/dev/null | head -1)"
done
for rr in ${RR:-3 4 5 6 7 8 10 14 15 16 29}; do
  echo "row ranges $rr: $(NVSR_WGRAD_ROW_RANGES=$rr NVSR_HIP_LIB=$R/scratch/variants/wg_0.so python3 $R/tools/conv_wgrad_time.py 2>/dev/null | head -1)"
done
