#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the unit-tensor kernels on the largest layer (tools/pmc_units.py): product library against tools/variants/<name>.so
# usage: tools/pmc_units_ab.sh <variant name>   (separate --pmc passes, kernel trace only)
VAR=${1:-noxcd}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in product $VAR; do
  lib=""; [ $v = $VAR ] && lib=$R/tools/variants/$VAR.so
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$v$C
    (cd $R && rocprofv3 --pmc $C --kernel-trace -d /tmp/pm_$v$C -o p --output-format csv -- python tools/pmc_units.py $lib > /tmp/pm_$v$C.log 2>&1)
    f=$(find /tmp/pm_$v$C -name "*counter_collection.csv" | head -1)
    python - "$f" "$v" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:50]
    if "units" not in k or "pack" in k: continue
    agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for (k, c), v in sorted(agg.items()):
    print(sys.argv[2], k, c, round(v / n[(k, c)]), "KiB/launch")
PY
  done
done
