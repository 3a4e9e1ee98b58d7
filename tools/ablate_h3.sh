#!/bin/bash
# Timing ablations of the fp16 two-part field kernels (csrc/field_h3.h H3_ABL_* hooks): builds side libraries
# nefes_amd/abl/libnefes_<name>.so that differ from the shipped one only in field_fwd_h3 / field_bwd_h3, then (on the GPU box,
# `tools/ablate_h3.sh run`) times the headline frame with each.  The ablated kernels compute garbage: timing only.
#   tools/ablate_h3.sh build     (CPU container)      tools/ablate_h3.sh run      (GPU box; prints one line per variant)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/nefes_amd/csrc
OUT=$ROOT/nefes_amd/abl
VARIANTS="base:-DH3_ABL_BASE cheapsplit:-DH3_ABL_CHEAPSPLIT nomask:-DH3_ABL_NOMASK norelu:-DH3_ABL_NORELU nomax3:-DH3_ABL_NOMAX3 allvalu:-DH3_ABL_CHEAPSPLIT,-DH3_ABL_NOMASK,-DH3_ABL_NORELU,-DH3_ABL_NOMAX3 allvalu_nodma:-DH3_ABL_CHEAPSPLIT,-DH3_ABL_NOMASK,-DH3_ABL_NORELU,-DH3_ABL_NOMAX3,-DH3_ABL_NODMA allvalu_nodma_nobias:-DH3_ABL_CHEAPSPLIT,-DH3_ABL_NOMASK,-DH3_ABL_NORELU,-DH3_ABL_NOMAX3,-DH3_ABL_NODMA,-DH3_ABL_NOBIAS"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -mllvm -pragma-unroll-threshold=65536"
if [ "$1" = "build" ]; then
  mkdir -p $OUT
  OTHERS=$(ls $CS/build/*.o | grep -v "field_fwd_h3\|field_bwd_h3")
  for v in $VARIANTS; do
    name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
    ( cd $CS
      /opt/rocm/bin/hipcc $FLAGS $defs -c field_fwd_h3.hip -o $OUT/fwd_$name.o &
      /opt/rocm/bin/hipcc $FLAGS $defs -DNEFES_TU_PART=1 -c field_fwd_h3.hip -o $OUT/fwd1_$name.o &
      /opt/rocm/bin/hipcc $FLAGS $defs -c field_bwd_h3.hip -o $OUT/bwd_$name.o &
      /opt/rocm/bin/hipcc $FLAGS $defs -DNEFES_TU_PART=1 -c field_bwd_h3.hip -o $OUT/bwd1_$name.o &
      wait
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libnefes_$name.so $OTHERS $OUT/fwd_$name.o $OUT/fwd1_$name.o $OUT/bwd_$name.o $OUT/bwd1_$name.o )
    rm -f $OUT/*_$name.o
    echo built $name
  done
else
  for v in $VARIANTS; do
    name=${v%%:*}
    NEFES_HIP_LIB=$OUT/libnefes_$name.so python $ROOT/bench.py --steps 2 --warmup 1 --cpu-rows 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('%-14s frame %.1f ms  fwd_full %.1f  bwd %.1f  sigma %.1f' % ('$name', d['ms_per_step'], k.get('field_fwd[full,h3]',0), k.get('field_bwd[h3]',0), k.get('field_fwd[sigma,h3]',0)))"
  done
fi
