#!/bin/bash
# Where the time of the SR weight-gradient contraction goes (run on the GPU box).  Variant libraries are built beforehand with
#   NVSR_EXTRA_HIPCC_FLAGS="-DWG_TUNE -DWG_ABLATE=<bits>" build_extension(out_path='scratch/variants/wg_<bits>.so')
# (sr_bwd.hip: 1 no global fetches, 2 no split + LDS writes, 4 no MFMAs -- wrong results by design); PIECES = workgroup counts to try.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for bits in ${BITS:-0 1 3 4}; do
  echo "WG_ABLATE=$bits: $(NVSR_HIP_LIB=$R/scratch/variants/wg_$bits.so python3 $R/tools/conv_wgrad_time.py $SHAPE 2>/dev/null | head -1)"
done
for n in ${PIECES:-128 256 384 512 768 1024}; do
  echo "pieces $n: $(NVSR_WGRAD_PIECES=$n NVSR_HIP_LIB=$R/scratch/variants/wg_0.so python3 $R/tools/conv_wgrad_time.py $SHAPE 2>/dev/null | head -1)"
done
