#!/bin/bash
# PMC passes of tools/time_h4.py (production sigma-only kernel next to the 16x16x32 experiment) -> gpurun_out/h4pmc.json
R=$(pwd); export TMPDIR=/tmp; OUT=$R/gpurun_out/h4pmc; mkdir -p $OUT; cd /tmp
i=1
for P in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/tools/time_h4.py > $OUT/p$i.log 2>&1
  i=$((i+1))
done
cd $R
python tools/pmc_aggregate.py gpurun_out/h4pmc.json $OUT/p1 $OUT/p2
tail -3 $OUT/p1.log
rm -rf $OUT
python - <<'PY'
import json
d = json.load(open("gpurun_out/h4pmc.json"))
for k, v in d.items():
    if "field" in k:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        print(k[:46], "cycles/CU %.0fk" % (cyc / 1e3), "MFMA busy %.1f%%" % (100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)),
              "wait %.1f%%" % (100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]), "VALU/MFMA %.2f" % (v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"]),
              "LDS inst/MFMA %.2f" % (v["SQ_INSTS_LDS"] / v["SQ_INSTS_MFMA"]), "LDS busy %.2f" % (v["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)),
              "bank conflicts %.1f%%" % (100 * v["SQ_LDS_BANK_CONFLICT"] / max(1, v["SQ_LDS_IDX_ACTIVE"])))
PY
