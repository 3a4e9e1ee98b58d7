#!/bin/bash
# A/B of a compile-time switch of the fp16 two-part field kernels on one box: a side library that differs from the shipped one
# only in field_fwd_h3 / field_bwd_h3 (parts 0, 2 and 4: the Wd = 256 and Wd = 128 frequency-embedding instances, the Wd = 128 train
# instances; AB_WORKLOADS="train" times the training step).
#   tools/ab_h3.sh build NAME "-DFLAG ..."   (CPU container)  ->  nefes_amd/abl/libnefes_NAME.so
#   tools/ab_h3.sh run NAME [NAME ...]       (GPU box)        ->  shipped library and each NAME, twice, headline + ref workloads
# Switches: -DNEFES_SINCOS_F64 (round-1 f64 argument reduction of the embedding: 453.4 vs 446.8 ms per headline frame).
# (Tried this way and dropped: requesting the next tile's ray inputs during the current tile's last layers -- no difference.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/nefes_amd/csrc; OUT=$ROOT/nefes_amd/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -mllvm -pragma-unroll-threshold=65536"
if [ "$1" = "build" ]; then
  NAME=$2; DEFS=$3; mkdir -p $OUT
  OTHERS=$(ls $CS/build/*.o | grep -v "field_fwd_h3.hip.o\|field_fwd_h3.p2.o\|field_fwd_h3.p4.o\|field_bwd_h3.hip.o\|field_bwd_h3.p2.o\|field_bwd_h3.p4.o")
  ( cd $CS
    /opt/rocm/bin/hipcc $FLAGS $DEFS -c field_fwd_h3.hip -o $OUT/f0_$NAME.o &
    /opt/rocm/bin/hipcc $FLAGS $DEFS -mllvm -amdgpu-mfma-vgpr-form -DNEFES_TU_PART=2 -c field_fwd_h3.hip -o $OUT/f2_$NAME.o &
    /opt/rocm/bin/hipcc $FLAGS $DEFS -DNEFES_H3_WIDE_MIN=99 -c field_bwd_h3.hip -o $OUT/b0_$NAME.o &
    /opt/rocm/bin/hipcc $FLAGS $DEFS -DNEFES_H3_WIDE_MIN=99 -mllvm -amdgpu-mfma-vgpr-form -DNEFES_TU_PART=2 -c field_bwd_h3.hip -o $OUT/b2_$NAME.o &
    /opt/rocm/bin/hipcc $FLAGS $DEFS -mllvm -amdgpu-mfma-vgpr-form -DNEFES_TU_PART=4 -c field_fwd_h3.hip -o $OUT/f4_$NAME.o &
    /opt/rocm/bin/hipcc $FLAGS $DEFS -DNEFES_H3_WIDE_MIN=99 -mllvm -amdgpu-mfma-vgpr-form -DNEFES_TU_PART=4 -c field_bwd_h3.hip -o $OUT/b4_$NAME.o &
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libnefes_$NAME.so $OTHERS $OUT/f0_$NAME.o $OUT/f2_$NAME.o $OUT/b0_$NAME.o $OUT/b2_$NAME.o $OUT/f4_$NAME.o $OUT/b4_$NAME.o )
  rm -f $OUT/f0_$NAME.o $OUT/f2_$NAME.o $OUT/b0_$NAME.o $OUT/b2_$NAME.o $OUT/f4_$NAME.o $OUT/b4_$NAME.o
  echo built $NAME
else
  shift
  for rep in 1 2; do for name in shipped "$@"; do
    lib=$OUT/libnefes_$name.so; [ $name = shipped ] && lib=$ROOT/nefes_amd/libnefes_hip.so
    for wl in ${AB_WORKLOADS:-metric ref}; do
      EXTRA="--steps 2 --warmup 1"; [ $wl = ref ] && EXTRA="--steps 50 --warmup 5"; [ $wl = train ] && EXTRA=""
      NEFES_HIP_LIB=$lib python $ROOT/bench.py --workload $wl $EXTRA --cpu-rows 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('%-14s %-7s step %.3f ms ' % ('$name', '$wl', d['ms_per_step']), {kk: round(v, 3) for kk, v in k.items() if 'field' in kk})"
    done
  done; done
fi
