#!/bin/bash
# Round profile (run on the GPU box through gpurun): bench lines + rocprofv3 kernel stats + the two HBM-counter passes.
# usage: tools/profile_round.sh <round-tag> [lines|pmc]     outputs under gpurun_out/<round-tag>/
#   lines: the bench lines + the kernel-stat runs; pmc: the counter passes; neither: both (a gpurun call is limited to 20 minutes: two calls)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=${1:-r01}; PART=${2:-all}; O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != pmc ]; then
python3 $R/bench.py --full-record $O/bench_render.json > $O/bench_render.line 2> $O/bench_render.err
python3 $R/bench.py --workload train --steps 30 --warmup 5 --full-record $O/bench_train.json > $O/bench_train.line 2> $O/bench_train.err
python3 $R/bench.py --workload train --train-what planes+decoder --steps 30 --warmup 5 --full-record $O/bench_train_dec.json > $O/bench_train_dec.line 2> $O/bench_train_dec.err
python3 $R/bench.py --workload sr --steps 3 --warmup 1 --full-record $O/bench_sr.json > $O/bench_sr.line 2> $O/bench_sr.err
python3 $R/bench.py --workload refine --refine-what joint --steps 8 --warmup 2 --full-record $O/bench_refine_joint.json > $O/bench_refine_joint.line 2> $O/bench_refine_joint.err
python3 $R/bench.py --workload refine --refine-what sr --steps 8 --warmup 2 --full-record $O/bench_refine_sr.json > $O/bench_refine_sr.line 2> $O/bench_refine_sr.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_render -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-modes > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 $R/bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_dec -- python3 $R/bench.py --workload train --train-what planes+decoder --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sr -- python3 $R/bench.py --workload sr --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_refine_joint -- python3 $R/bench.py --workload refine --refine-what joint --steps 5 --warmup 1 --no-cpu-baseline --full-record $O/stats_refine_joint.json > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_refine_sr -- python3 $R/bench.py --workload refine --refine-what sr --steps 5 --warmup 1 --no-cpu-baseline --full-record $O/stats_refine_sr.json > /dev/null 2>&1
find $O/stats_* -name "*kernel_trace.csv" -delete
fi
if [ "$PART" != lines ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-modes > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_train_$c -- python3 $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_train_dec_$c -- python3 $R/bench.py --workload train --train-what planes+decoder --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_sr_$c -- python3 $R/bench.py --workload sr --steps 1 --warmup 0 --no-cpu-baseline --no-modes > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_refine_joint_$c -- python3 $R/bench.py --workload refine --refine-what joint --steps 1 --warmup 1 --no-cpu-baseline --no-split > $O/pmc_refine_joint_$c.json 2> /dev/null
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_refine_sr_$c -- python3 $R/bench.py --workload refine --refine-what sr --steps 1 --warmup 1 --no-cpu-baseline --no-split > $O/pmc_refine_sr_$c.json 2> /dev/null
done
# matrix-pipe utilisation and clock of the fine pass (one more counter pass of the same command)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-modes > /dev/null 2>&1
# executed matrix instructions of the SR stage and of a refine iteration (executed / algorithmic: tile rounding on ragged crops)
rocprofv3 --pmc SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_sr_SQ_INSTS_MFMA -- python3 $R/bench.py --workload sr --steps 1 --warmup 0 --no-cpu-baseline --no-modes > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_refine_joint_SQ_INSTS_MFMA -- python3 $R/bench.py --workload refine --refine-what joint --steps 1 --warmup 1 --no-cpu-baseline --no-split > /dev/null 2>&1
# issue counters of the training kernels (fine-pass forward / gate-driven backward): tools/train_pmc.sh (4 counter passes of the train bench)
bash $R/tools/train_pmc.sh > $O/train_issue_counters.txt 2>&1
# keep the merged-back payload small (gpurun copies back at most 64 MiB): the counter passes' kernel traces are not read by tools/profile_collect.py
find $O/pmc_* $R/gpurun_out/pmc_t* -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $O/* $R/gpurun_out/pmc_t* 2>/dev/null | sort -h | tail -8
fi
[ -f $O/bench_render.line ] && tail -c 600 $O/bench_render.line
