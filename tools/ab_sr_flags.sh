#!/bin/bash
# same-box A/B of compile-time switches of the SR files on the SR workloads (generalises tools/ab_reduce_lane.sh):
#   tools/ab_sr_flags.sh build "<flags>" "<files, e.g. sr.hip sr_bwd.hip>" <name>     (here: hipcc cross-compiles) -> scratch/variants/ab_<name>.so
#   tools/ab_sr_flags.sh run <name> [<name> ...]                                        (GPU box) product and variants alternate, two rounds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; V=$R/scratch/variants; mkdir -p $V $R/gpurun_out
if [ "$1" = build ]; then
  NVSR_VARIANT_ONLY="$3" NVSR_EXTRA_HIPCC_FLAGS="$2" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/ab_$4.so')" > $V/ab_$4.log 2>&1 && echo "built ab_$4 ($2 on $3)" || tail -5 $V/ab_$4.log
  exit 0
fi
shift; cd $R
line() { python3 -c "import json,sys; r=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print('%.2f ms (%.3f of roof)' % (r['ms_per_step'], r['roofline']['frac']) if r.get('roofline') else '%.2f ms' % r['ms_per_step'])" $1 2>/dev/null || echo "n/a"; }
for round in 1 2; do
  for lib in product "$@"; do
    if [ $lib = product ]; then unset NVSR_HIP_LIB; else export NVSR_HIP_LIB=$V/ab_$lib.so; fi
    timeout -k 10 200 python3 bench.py --workload sr --steps 4 --warmup 2 --no-cpu-baseline --no-modes > gpurun_out/ab_sr_${lib}_$round.log 2>/dev/null
    timeout -k 10 200 python3 bench.py --workload refine --refine-what sr --steps 10 --warmup 3 --no-split --no-cpu-baseline > gpurun_out/ab_ref_${lib}_$round.log 2>/dev/null
    echo "round $round $lib: sr $(line gpurun_out/ab_sr_${lib}_$round.log) | refine (what = SR) $(line gpurun_out/ab_ref_${lib}_$round.log)"
  done
done
