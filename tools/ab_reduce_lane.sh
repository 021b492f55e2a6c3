set -e
R=$GRAFT_REPO_ROOT; V=$R/gpurun_out/variants; mkdir -p $V
NVSR_VARIANT_ONLY=sr_bwd.hip NVSR_EXTRA_HIPCC_FLAGS="-DWG_REDUCE_LANE=0" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/nolane.so')" > $R/gpurun_out/ab_build.log 2>&1
for round in 1 2; do
  for lib in lane nolane; do
    if [ $lib = nolane ]; then export NVSR_HIP_LIB=$V/nolane.so; else unset NVSR_HIP_LIB; fi
    timeout -k 10 200 python3 bench.py --workload sr --steps 4 --warmup 2 > gpurun_out/ab_sr_${lib}_$round.log 2>&1
    timeout -k 10 150 python3 bench.py --workload refine --refine-what sr --steps 10 --warmup 3 --no-split > gpurun_out/ab_ref_${lib}_$round.log 2>&1
    echo "$lib $round done"
  done
done
