#!/bin/bash
# Round profile (run on the GPU box through gpurun): bench lines + rocprofv3 kernel stats + the two HBM-counter passes.
# usage: tools/profile_round.sh <round-tag>      outputs under gpurun_out/<round-tag>/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=${1:-r01}; O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_render.json 2> $O/bench_render.err
python3 $R/bench.py --workload train --steps 30 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err
python3 $R/bench.py --workload train --train-what planes+decoder --steps 30 --warmup 5 > $O/bench_train_dec.json 2> $O/bench_train_dec.err
python3 $R/bench.py --workload sr --steps 3 --warmup 1 > $O/bench_sr.json 2> $O/bench_sr.err
python3 $R/bench.py --workload refine --refine-what joint --steps 8 --warmup 2 > $O/bench_refine_joint.json 2> $O/bench_refine_joint.err
python3 $R/bench.py --workload refine --refine-what sr --steps 8 --warmup 2 > $O/bench_refine_sr.json 2> $O/bench_refine_sr.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_render -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-modes > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 $R/bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_dec -- python3 $R/bench.py --workload train --train-what planes+decoder --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sr -- python3 $R/bench.py --workload sr --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_refine_joint -- python3 $R/bench.py --workload refine --refine-what joint --steps 5 --warmup 1 --no-cpu-baseline > $O/stats_refine_joint.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_refine_sr -- python3 $R/bench.py --workload refine --refine-what sr --steps 5 --warmup 1 --no-cpu-baseline > $O/stats_refine_sr.json 2> /dev/null
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
# issue counters of the training kernels (fine-pass forward / gate-driven backward): tools/train_pmc.sh (4 counter passes of the train bench)
bash $R/tools/train_pmc.sh > $O/train_issue_counters.txt 2>&1
# keep the merged-back payload small: drop the per-dispatch traces of the stats runs (the *_kernel_stats.csv summaries stay)
find $O/stats_* -name "*kernel_trace.csv" -delete
tail -c 400 $O/bench_render.json
