#!/bin/bash
# Copies what tools/round_collect.sh left under gpurun_out/<round> into profiles/<round> (the tracked copies the README tables cite):
#   bash tools/install_collection.sh r05
set -e
R=${1:?round, e.g. r05}
S=gpurun_out/$R; D=profiles/$R
mkdir -p $D/secondary
cp $S/bench_full.json $S/bench_full_kernel_stats.csv $S/bench_full_kernel_trace_nefes.csv $S/pmc_pass1_sq.csv $S/pmc_pass2_fetch.csv \
   $S/pmc_pass3_write_lds.csv $S/pmc_per_launch.json $D/
for w in ref cam loop50 train metric128 metric_f32 cam_unfused; do [ -f $S/bench_$w.json ] && cp $S/bench_$w.json $D/secondary/; done
for w in ref cam loop50 train metric128 loop1; do [ -f $S/${w}_kernel_stats.csv ] && cp $S/${w}_kernel_stats.csv $D/secondary/; done
python tools/collect_parity.py ${R#r}
python - <<PY
import json
d = json.load(open("$D/bench_full.json"))
print("headline", round(d["ms_per_step"], 2), "ms", round(d["value"]), "rays/s frac", round(d["roofline"]["frac"], 4), d["roofline"]["traffic_source"])
PY
