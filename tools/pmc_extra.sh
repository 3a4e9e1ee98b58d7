#!/bin/bash
# Extra PMC passes (instruction cache, wait breakdown) on a reduced frame: tools/pmc_extra.sh -> gpurun_out/pmcx.json
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcx; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=1
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY" \
         "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --height 120 --steps 1 --warmup 0 --cpu-rows 0 > $OUT/p$i.log 2>&1
  i=$((i+1))
done
cd $ROOT
python tools/pmc_aggregate.py gpurun_out/pmcx.json $OUT/p1 $OUT/p2 $OUT/p3
rm -rf $OUT
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmcx.json'))
for k,v in d.items():
    if 'field' in k:
        print(k); print('  ', {c: f"{x:.4g}" for c,x in v.items()})
PY
