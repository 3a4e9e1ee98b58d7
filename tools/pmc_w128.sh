#!/bin/bash
# Where the Wd = 128 field kernels' cycles go: three PMC passes over `bench.py --workload ref` -- what the waves are
# busy with (SQ_ACTIVE_INST_*), what they wait for (SQ_WAIT_*, SQ_INST_LEVEL_*), and the LDS front end (FIFO full, conflicts).
#   tools/pmc_w128.sh r05 [NEFES_HIP_LIB]   ->  gpurun_out/<round>/w128_pmc_per_launch.json
R=${1:-r05}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp; [ -n "$2" ] && export NEFES_HIP_LIB=$2; cd /tmp
i=1
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_INSTS_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_SMEM"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/w128pmc$i -- python3 $ROOT/bench.py --workload ref --steps 10 --warmup 2 --cpu-rows 0 > $OUT/w128pmc$i.log 2>&1
  rc=$?
  # a pass that failed, timed out or produced no counter file must not be aggregated silently (eight counters may not fit one pass on
  # every firmware): say so, keep its log and directory, and stop -- a digest with counters missing looks like a digest
  if [ $rc -ne 0 ] || [ -z "$(find $OUT/w128pmc$i -name '*counter_collection.csv' 2>/dev/null | head -1)" ]; then
    echo "pmc_w128.sh: pass $i (rc $rc) gave no counter file: see $OUT/w128pmc$i.log; nothing aggregated, directories kept" >&2
    tail -5 $OUT/w128pmc$i.log >&2
    exit 1
  fi
  i=$((i+1))
done
cd $ROOT
python tools/pmc_aggregate.py $OUT/w128_pmc_per_launch.json $OUT/w128pmc1 $OUT/w128pmc2 $OUT/w128pmc3 $OUT/w128pmc4 || { echo "pmc_w128.sh: aggregation failed, pass directories kept" >&2; exit 1; }
tail -2 $OUT/w128pmc4.log
rm -rf $OUT/w128pmc1 $OUT/w128pmc2 $OUT/w128pmc3 $OUT/w128pmc4
python - <<PY
import json
d = json.load(open("$OUT/w128_pmc_per_launch.json"))
for k, v in d.items():
    if "field" not in k:
        continue
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for n in sorted(v):
        if n.startswith("_"): continue
        print("   %-28s %14.0f   %.3f of SQ_WAVE_CYCLES" % (n, v[n], v[n] / wc))
PY
