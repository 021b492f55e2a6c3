#!/bin/bash
# Race probe (nvsr_common.h NVSR_RACE_PROBE): a probe library in which every wave but the first of every LDS-filling kernel starts
# ~30 us late, and the GPU parity tests run against it through NVSR_HIP_LIB -- never the product library.
#   tools/race_probe.sh build     (here: hipcc cross-compiles)  -> scratch/variants/race_probe.so  + race_probe_nobarrier.so (negative control: the
#                                  round-5 race re-opened with -DBL_PROLOGUE_BARRIER=0; the probe must make the backward tests FAIL on it)
#   tools/race_probe.sh run       (GPU box) -> gpurun_out/race_probe_*.log
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; V=$R/scratch/variants; mkdir -p $V $R/gpurun_out
case "${1:-run}" in
build)
  NVSR_EXTRA_HIPCC_FLAGS="-DNVSR_RACE_PROBE=1" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/race_probe.so')" > $V/race_probe.log 2>&1 && echo built race_probe.so || tail -5 $V/race_probe.log
  NVSR_VARIANT_ONLY="render_bwd_limb.hip" NVSR_EXTRA_HIPCC_FLAGS="-DNVSR_RACE_PROBE=1 -DBL_PROLOGUE_BARRIER=0" python3 -c "import sys; sys.path.insert(0, '$R'); import nvsr_amd; nvsr_amd.build_extension(out_path='$V/race_probe_nobarrier.so')" > $V/race_probe_nobarrier.log 2>&1 && echo built race_probe_nobarrier.so || tail -5 $V/race_probe_nobarrier.log
  ;;
run)
  cd $R
  # negative control first: the known race, re-opened, must be caught by the gradient tests
  NVSR_HIP_LIB=$V/race_probe_nobarrier.so timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py -q -m gpu -k "plane_gradients or grads or f16_backward" -p no:cacheprovider > gpurun_out/race_probe_negative_control.log 2>&1
  echo "negative control (barrier removed, probe on): $(tail -1 gpurun_out/race_probe_negative_control.log)"
  # the probe proper: every GPU test (the oracle / golden parity tests among them) on the probe library
  NVSR_HIP_LIB=$V/race_probe.so timeout -k 10 1500 python3 -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_host.py > gpurun_out/race_probe_suite.log 2>&1
  echo "probe suite: $(tail -1 gpurun_out/race_probe_suite.log)"
  ;;
esac
