#!/bin/bash
# HBM-side traffic of the training step's kernels (bench.py --workload train), per launch: FETCH_SIZE / WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md: units of 32 B on gfx950... see tools/pmc_aggregate.py users) + the stats pass for the durations:
#   tools/profile_train_pmc.sh r02  ->  gpurun_out/<round>/train_pmc_per_launch.json (+ a digest: GB per launch, TB/s)
R=${1:-r02}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=1
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/trpmc$i -- python3 $ROOT/bench.py --workload train --cpu-rows 0 > $OUT/trpmc$i.log 2>&1
  i=$((i+1))
done
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trst -- python3 $ROOT/bench.py --workload train --cpu-rows 0 > $OUT/trst.log 2>&1)
cd $ROOT
python tools/pmc_aggregate.py $OUT/train_pmc_per_launch.json $OUT/trpmc1 $OUT/trpmc2
S=$(find $OUT/trst -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && head -25 $S > $OUT/train_kernel_stats.csv
tail -2 $OUT/trpmc2.log | cut -c1-300
rm -rf $OUT/trpmc1 $OUT/trpmc2 $OUT/trst $OUT/trpmc1.log $OUT/trpmc2.log $OUT/trst.log
python - <<PY
import json, csv, re
d = json.load(open("$OUT/train_pmc_per_launch.json"))
def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)
st = {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open("$OUT/train_kernel_stats.csv"))}
for k, v in d.items():
    t = st.get(k)
    if t is None:
        continue
    # FETCH_SIZE / WRITE_SIZE: kilobytes on this counter set (rocprofv3 derived metrics), x2 on FETCH for gfx950 (profiles/r02/README)
    rd, wr = 2 * v.get("FETCH_SIZE", 0) * 1024 / 1e9, v.get("WRITE_SIZE", 0) * 1024 / 1e9
    print(f"{k:60s} {t:8.3f} ms  read {rd:7.2f} GB  write {wr:7.2f} GB  -> {(rd + wr) / t:6.2f} TB/s")
PY
