#!/bin/bash
# A/B of a compile-time switch of the fp16 two-part field kernels on one box: a side library that differs from the shipped one
# only in the objects of field_fwd_h3.hip / field_bwd_h3.hip (all parts; AB_WORKLOADS="train" times the training step).
#   tools/ab_h3.sh build NAME "-DFLAG ..." ["-DFLAG for the Wd = 128 objects only"]   (CPU container)  ->  nefes_amd/abl/libnefes_NAME.so
#   tools/ab_h3.sh run NAME [NAME ...]       (GPU box)        ->  shipped library and each NAME, twice, headline + ref workloads
# Switches: -DNEFES_SINCOS_F64 (round-1 f64 argument reduction of the embedding: 453.4 vs 446.8 ms per headline frame).
# (Tried this way and dropped: requesting the next tile's ray inputs during the current tile's last layers -- no difference.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/nefes_amd/csrc; OUT=$ROOT/nefes_amd/abl
if [ "$1" = "build" ]; then
  # the shipped objects are copied, the fp16 field kernels' are rebuilt by the library's own Makefile with the extra flags
  NAME=$2; DEFS=$3; DEFS128=$4; mkdir -p $OUT/$NAME
  cp -p $CS/build/*.o $OUT/$NAME/ && rm -f $OUT/$NAME/field_fwd_h3.* $OUT/$NAME/field_bwd_h3.*
  make -C $CS -j${JOBS:-7} BUILD=../abl/$NAME OUT=../abl/libnefes_$NAME.so EXTRA_H3="$DEFS" EXTRA_W128="$DEFS128" > $OUT/$NAME.log 2>&1 || { tail -20 $OUT/$NAME.log; exit 1; }
  rm -rf $OUT/$NAME
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
import hashlib
g=hashlib.sha256(repr(d.get('pose_grad')).encode()).hexdigest()[:8]     # equal digests: the builds agree on all twelve numbers bit for bit
print('%-14s %-7s step %.3f ms ' % ('$name', '$wl', d['ms_per_step']), {kk: round(v, 3) for kk, v in k.items() if 'field' in kk}, 'pose_grad', g)"
    done
  done; done
fi
