#!/bin/bash
# PMC passes over tools/pmc_units.py (unit-tensor kernels, largest layer); summaries -> gpurun_out/pmc_units_*.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcu$i
  (cd $R && rocprofv3 --pmc $C --kernel-trace -d /tmp/pmcu$i -o p --output-format csv -- python tools/pmc_units.py > /tmp/pmcu$i.log 2>&1)
  f=$(find /tmp/pmcu$i -name "*counter_collection.csv" | head -1)
  python - "$f" > $R/gpurun_out/pmc_units_$i.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if "units" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k, {c: round(v / n[(k, c)]) for c, v in agg[k].items()})
PY
done
cat $R/gpurun_out/pmc_units_*.txt
