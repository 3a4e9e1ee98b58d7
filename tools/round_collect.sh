#!/bin/bash
# One gpurun call for the judged set of a round:  tools/round_collect.sh r03
# -m gpu tests (parity log), profile_round + profile_secondary, then the default bench line re-run with the fresh PMC file installed
# (bench.py quotes roofline.traffic only from a profile whose kernel-source digest matches the tree).
R=${1:-r04}
ROOT=$(pwd)      # run from the repository root (gpurun does); every path below is relative to it
mkdir -p gpurun_out/$R
export NEFES_PARITY_LOG=$PWD/gpurun_out/parity.jsonl
rm -f $NEFES_PARITY_LOG
python -m pytest tests -q -m gpu > gpurun_out/${R}_gputest.log 2>&1; tail -3 gpurun_out/${R}_gputest.log
bash tools/profile_round.sh $R > gpurun_out/${R}_profile.log 2>&1
bash tools/profile_secondary.sh $R > gpurun_out/${R}_secondary.log 2>&1
mkdir -p profiles/$R && cp gpurun_out/$R/pmc_per_launch.json profiles/$R/
python bench.py > gpurun_out/$R/bench_full.json 2> gpurun_out/$R/bench_full.err; cat gpurun_out/$R/bench_full.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$R/loop1 -- python3 $ROOT/tools/prof_loop1.py > $ROOT/gpurun_out/$R/loop1.log 2>&1
cd $ROOT
S=$(find gpurun_out/$R/loop1 -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S gpurun_out/$R/loop1_kernel_stats.csv
rm -rf gpurun_out/$R/loop1; tail -2 gpurun_out/$R/loop1.log
