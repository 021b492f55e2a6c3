#!/bin/bash
# A/B of compile-time switches on ONE box (boxes of the pool differ by several per cent): tools/ab_flags.sh "<flags A>" "<flags B>" [what]
# rebuilds the library with each flag set (NVSR_EXTRA_HIPCC_FLAGS) and times the train step (what = planes | planes+decoder | both), twice.
R=$GRAFT_REPO_ROOT; WHAT=${3:-both}
for round in 1 2; do
  for flags in "$1" "$2"; do
    NVSR_EXTRA_HIPCC_FLAGS="$flags" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(force=True)" > /dev/null 2>&1
    for w in planes planes+decoder; do
      if [ "$WHAT" = both ] || [ "$WHAT" = "$w" ]; then echo "[$flags] $(python3 $R/tools/train_step_time.py $w 2>/dev/null | head -1 | cut -c1-95)"; fi
    done
  done
done
python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(force=True)" > /dev/null 2>&1
