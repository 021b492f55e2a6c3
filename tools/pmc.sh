#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>   (runs tools/$PMC_SCRIPT $PMC_ARGS, default kbench_fine_pass.py 1, under rocprofv3 --pmc; a pass that does not finish in 150 s is killed)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=$1; shift
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/${PMC_SCRIPT:-kbench_fine_pass.py} ${PMC_ARGS-1} > $R/gpurun_out/pmc_$tag/out.log 2> $R/gpurun_out/pmc_$tag/err.log
echo "pass $tag rc=$? $(tail -1 $R/gpurun_out/pmc_$tag/out.log)"
