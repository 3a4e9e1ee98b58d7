#!/bin/bash
# PMC passes at the reference's refinement shape (80x60, 64+64 samples, 8x128, C=128: bench.py --workload ref), VERDICT r1 item 8:
#   tools/profile_ref_pmc.sh r02  ->  gpurun_out/<round>/ref_pmc_per_launch.json (+ a one-line digest per field kernel)
R=${1:-r02}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=1
for P in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/refpmc$i -- python3 $ROOT/bench.py --workload ref --steps 10 --warmup 2 --cpu-rows 0 > $OUT/refpmc$i.log 2>&1
  i=$((i+1))
done
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/refst -- python3 $ROOT/bench.py --workload ref --steps 50 --warmup 5 --cpu-rows 0 > $OUT/refst.log 2>&1)
cd $ROOT
python tools/pmc_aggregate.py $OUT/ref_pmc_per_launch.json $OUT/refpmc1 $OUT/refpmc2
S=$(find $OUT/refst -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && head -12 $S > $OUT/ref_kernel_stats_50steps.csv
tail -3 $OUT/refpmc2.log
rm -rf $OUT/refpmc1 $OUT/refpmc2 $OUT/refst $OUT/refpmc1.log $OUT/refpmc2.log $OUT/refst.log
python - <<PY
import json, csv
d = json.load(open("$OUT/ref_pmc_per_launch.json"))
st = {r["Name"]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open("$OUT/ref_kernel_stats_50steps.csv"))}
for k, v in d.items():
    if "field" not in k:
        continue
    t = next((x for n, x in st.items() if k.replace(" ", "") in n.replace(" ", "")), None)
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, "ms", t, "MFMA busy %.1f %%" % (100 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1, cyc * 1024)),
          "wait %.1f %%" % (100 * v.get("SQ_WAIT_ANY", 0) / max(1, v.get("SQ_WAVE_CYCLES", 1))),
          "VALU/MFMA %.2f" % (v.get("SQ_INSTS_VALU", 0) / max(1, v.get("SQ_INSTS_MFMA", 1))),
          "LDS/MFMA %.2f" % (v.get("SQ_INSTS_LDS", 0) / max(1, v.get("SQ_INSTS_MFMA", 1))))
PY
