#!/bin/bash
# In-kernel cycle stamps of the fp16 forward kernels (csrc/field_h3.h H3_STAMP): cycles spent inside the wide (8-tile) product
# runs and inside the ring acquires, per MFMA.  tools/stamp_h3.sh build (CPU container) / run (GPU box).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/nefes_amd/csrc; OUT=$ROOT/nefes_amd/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -mllvm -pragma-unroll-threshold=65536 -DH3_STAMP"
VARIANTS="base: nosplit:-DH3_ABL_NOSPLIT mfmaonly:-DH3_ABL_NOSPLIT,-DH3_ABL_NOSTORE,-DH3_ABL_NOLOAD,-DH3_ABL_NOAREAD,-DH3_ABL_NOBIAS"
if [ "$1" = "build" ]; then
  mkdir -p $OUT
  # STAMP_W=128: the Wd=128 instances live in the third translation unit of field_fwd_h3.hip (NEFES_TU_PART=2, VGPR-form MFMAs)
  if [ "${STAMP_W:-256}" = "128" ]; then SKIP="field_fwd_h3.p2.o"; PART="-DNEFES_TU_PART=2 -mllvm -amdgpu-mfma-vgpr-form"; else SKIP="field_fwd_h3.hip.o"; PART=""; fi
  OTHERS=$(ls $CS/build/*.o | grep -v "$SKIP")
  for v in $VARIANTS; do
    name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
    ( cd $CS && /opt/rocm/bin/hipcc $FLAGS -DH3_STAMP_READER $PART $defs -c field_fwd_h3.hip -o $OUT/fwd_stamp_$name.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libnefes_stamp_$name.so $OTHERS $OUT/fwd_stamp_$name.o ) &
  done
  wait
  rm -f $OUT/*.o
  echo built
else
 for v in $VARIANTS; do
  name=${v%%:*}
  echo "== $name"
  NEFES_HIP_LIB=$OUT/libnefes_stamp_$name.so python - <<'PY'
import ctypes, sys, types, torch
sys.path.insert(0, '.')
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF
lib = L.load()
raw = ctypes.CDLL(L.LIB_PATH)
import os
WD = int(os.environ.get('STAMP_W', '256')); CF = 16 if WD == 256 else 128
net = NeRFH_NFF('coarse', W=WD, f_dim=CF).requires_grad_(False).cuda()
pk = net.packed()
N, S = 76800, 64
g = torch.Generator().manual_seed(0)
o = (torch.randn(N, 3, generator=g) * 0.3).cuda(); d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda()
z = torch.sort(torch.rand(N, S, generator=g) * 4, -1)[0].cuda()
out = (ctypes.c_ulonglong * 4)()
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.field_fwd_x6(pk, L.FIELD_SIGMA, N, S, o, d, z)
    e1.record()
    torch.cuda.synchronize()
    raw.nefes_debug_h3_stamps(out)
run, acq, n, total = out[0], out[1], out[2], out[3]
ms = e0.elapsed_time(e1)
cus = torch.cuda.get_device_properties(0).multi_processor_count
print(f"{run / n:.1f} cycles per MFMA inside the wide runs, of which {acq / n:.1f} in ring acquires (wait + barrier); "
      f"wide runs = {100.0 * run / total:.1f} % of the kernel's cycles; {total / cus / 1e3:.0f} k cycles per CU in {ms:.2f} ms = {total / cus / ms / 1e6:.2f} GHz stamp clock")
PY
 done
fi
