#!/bin/bash
# PMC passes over csrc/conv.hip at FusionNet's layer shapes (tools/time_conv.py):  tools/pmc_conv.sh  -> gpurun_out/conv_pmc.json
ROOT=$(pwd); OUT=$ROOT/gpurun_out/convpmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
P1="GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/time_conv.py > $OUT/p$i.log 2>&1
  i=$((i+1))
done
cd $ROOT
python tools/pmc_aggregate.py $OUT/conv_pmc.json $OUT/p1 $OUT/p2 $OUT/p3
python - <<'PY'
import json
j=json.load(open('gpurun_out/convpmc/conv_pmc.json'))
for k,v in j.items():
    if 'conv2d' in k: print(k, json.dumps(v))
PY
tail -14 $OUT/p1.log
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
