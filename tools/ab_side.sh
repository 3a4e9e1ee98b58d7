#!/bin/bash
# A/B of side libraries on one box: bench.py --workload $W, alternating
W=${1:-ref}; shift
out=gpurun_out/ab_$W.log; : > $out
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = main ]; then unset NEFES_HIP_LIB; else export NEFES_HIP_LIB=$PWD/nefes_amd/side/$lib; fi
    python bench.py --workload $W --cpu-rows 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', '$W', round(d.get('ms_per_step') or d.get('ms_per_image_50_iterations'), 4), json.dumps(d.get('kernels_ms')))" >> $out
  done
done
cat $out
