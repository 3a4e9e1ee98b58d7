#!/bin/bash
# What the train-mode forward / dX kernels WAIT for (round 6): two PMC passes over `bench.py --workload train`.
#   tools/pmc_train_wait.sh r06 [NEFES_HIP_LIB]  ->  gpurun_out/<round>/train_wait_pmc.json
R=${1:-r06}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp; [ -n "$2" ] && export NEFES_HIP_LIB=$2; cd /tmp
i=1
# (one pass of SQ counters.  A second pass of TA_* / TCC_*_sum counters did not finish in 600 s on this workload -- twice -- and is not tried again)
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/twpmc$i -- python3 $ROOT/bench.py --workload train --cpu-rows 0 > $OUT/twpmc$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ] || [ -z "$(find $OUT/twpmc$i -name '*counter_collection.csv' 2>/dev/null | head -1)" ]; then
    echo "pmc_train_wait.sh: pass $i (rc $rc) gave no counter file" >&2; tail -5 $OUT/twpmc$i.log >&2; exit 1
  fi
  i=$((i+1))
done
cd $ROOT
python tools/pmc_aggregate.py $OUT/train_wait_pmc${3:-}.json $OUT/twpmc1
rm -rf $OUT/twpmc1
python - <<PY
import json
d = json.load(open("$OUT/train_wait_pmc${3:-}.json"))
for k, v in d.items():
    if "field_" not in k: continue
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print(k[:90])
    for n in sorted(v):
        if n.startswith("_"): continue
        print("   %-36s %16.0f   %.3f of SQ_WAVE_CYCLES" % (n, v[n], v[n] / wc))
PY
