#!/bin/bash
# rocprofv3 kernel stats of the secondary workloads:  tools/profile_secondary.sh r01  -> gpurun_out/<round>/<workload>_*
R=${1:-r01}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp
for wl in ref cam loop50 train metric128; do
  EXTRA=""; [ $wl = ref ] && EXTRA="--steps 50 --warmup 5"     # a 2 ms step: three steps after one warm-up still carry one-off host work
  python bench.py --workload $wl --cpu-rows 0 $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$wl.json
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$wl -- python3 $ROOT/bench.py --workload $wl --cpu-rows 0 > $OUT/st_$wl.log 2>&1)
  S=$(find $OUT/st_$wl -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && head -25 $S > $OUT/${wl}_kernel_stats.csv
  rm -rf $OUT/st_$wl $OUT/st_$wl.log
done
# the strict-fp32 anchor on the current sources: the headline frame on the fp32-MFMA kernels (v_mfma_f32_32x32x2_f32,
# exact fp32 FMA chain: what "x the fp32 MFMA pipe" in README / DESIGN is measured against), and the hash-grid frame without the
# in-kernel gathers (NEFES_FUSED_HASHGRID=0: the launches section 4.8 replaced)
NEFES_SPLIT=f32 python bench.py --cpu-rows 0 --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_metric_f32.json
NEFES_FUSED_HASHGRID=0 python bench.py --workload cam --cpu-rows 0 2>/dev/null | tail -1 > $OUT/bench_cam_unfused.json
cut -c1-400 $OUT/bench_*.json
