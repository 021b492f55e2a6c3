#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>   (runs kbench under rocprofv3 --pmc)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=$1; shift
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/kbench_fine_pass.py 1 > $R/gpurun_out/pmc_$tag/out.log 2> $R/gpurun_out/pmc_$tag/err.log
tail -1 $R/gpurun_out/pmc_$tag/out.log
