#!/bin/bash
# Collect the judged profile set of one round on the GPU box:  tools/profile_round.sh r01
# Writes gpurun_out/<round>/ (copy what is to be judged into profiles/<round>/ afterwards).
# rocprofv3 is run from /tmp with TMPDIR=/tmp, the program directly after `--`, PMC passes separate from the stats pass.
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-rows 0 > $OUT/stats.log 2>&1
P1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="FETCH_SIZE"
P3="WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout 900 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-rows 0 > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
cd $ROOT
python tools/pmc_aggregate.py $OUT/pmc_per_launch.json $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/bench_full_kernel_stats.csv
T=$(find $OUT/stats -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && grep -E "Kernel_Name|nefes|_kernel" $T > $OUT/bench_full_kernel_trace_nefes.csv
i=1
for n in sq fetch write_lds; do
  C=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1); [ -n "$C" ] && grep -E "Counter_Name|_kernel" $C > $OUT/pmc_pass${i}_$n.csv
  i=$((i+1))
done
rm -rf $OUT/stats $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
cat $OUT/bench_full.json
head -12 $OUT/bench_full_kernel_stats.csv
