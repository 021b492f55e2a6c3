#!/bin/bash
# A/B of compile-time switches on ONE box (boxes of the pool differ by several per cent): tools/ab_flags.sh "<flags A>" "<flags B>" [what]
# builds a VARIANT library per flag set (never over the product library: nvsr_amd.build_extension(out_path=...), loaded through
# NVSR_HIP_LIB) and times the train step (what = planes | planes+decoder | both), twice.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; WHAT=${3:-both}
V=$R/gpurun_out/variants; mkdir -p $V
i=0
for flags in "$1" "$2"; do
  i=$((i+1))
  NVSR_EXTRA_HIPCC_FLAGS="$flags" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/ab_$i.so')" > /dev/null 2>&1
done
for round in 1 2; do
  i=0
  for flags in "$1" "$2"; do
    i=$((i+1))
    for w in planes planes+decoder; do
      if [ "$WHAT" = both ] || [ "$WHAT" = "$w" ]; then echo "[$flags] $(NVSR_HIP_LIB=$V/ab_$i.so python3 $R/tools/train_step_time.py $w 2>/dev/null | head -1 | cut -c1-95)"; fi
    done
  done
done
