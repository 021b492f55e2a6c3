#!/bin/bash
# Variant libraries for the attribution table of the planes-only backward (profiles/r06_bwd_attribution.txt): built HERE (hipcc
# cross-compiles), under scratch/variants/ (travels with the gpurun snapshot), only render_bwd_limb.hip recompiled per variant.
# name = BL_ABLATE bits [+ "s" = no limb splits at all (R3_ABLATE=4096 in limb_core.h)]; wrong results by design, never the product library.
R=$(cd "$(dirname "$0")/.." && pwd); V=$R/scratch/variants; mkdir -p $V
export NVSR_VARIANT_ONLY="render_bwd_limb.hip"
for v in ${VARIANTS:-0 4 6 14s 15s 79s 111s 1 64 65 97}; do
  bits=${v%s}; extra="-DBL_ABLATE=$bits"; [ "$v" != "$bits" ] && extra="$extra -DR3_ABLATE=4096"
  NVSR_EXTRA_HIPCC_FLAGS="$extra $EXTRA" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/bl_$v$TAG.so')" > $V/bl_$v$TAG.log 2>&1 \
    && echo "built bl_$v$TAG ($extra $EXTRA)" || { echo "bl_$v$TAG: build failed"; tail -5 $V/bl_$v$TAG.log; }
done
