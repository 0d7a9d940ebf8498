#!/bin/bash
# HBM traffic of the conv kernels in the roofline pass: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md: they do not fit one pass; FETCH_SIZE x2 on gfx950 for wide coalesced reads).
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${TAG}_$c -- python bench.py --roofline-only --steps 2 --warmup 1 > /tmp/pmc_${TAG}_$c.log 2>&1) || tail -3 /tmp/pmc_${TAG}_$c.log
done
cd $R && python - "$TAG" <<'PY' > $R/gpurun_out/${1:-rXX}_pmc_traffic.txt
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
print("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --roofline-only --steps 2 --warmup 1")
print("# per launch averages; counter unit = KiB.  hbm_MB = (2 x FETCH_SIZE + WRITE_SIZE) KiB / 1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads)")
print(f"{'kernel':64s} {'launches':>8s} {'FETCH_KiB':>12s} {'WRITE_KiB':>12s} {'hbm_MB':>9s}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    n = cnt[(k, "FETCH_SIZE")] or 1
    f = v.get("FETCH_SIZE", 0) / n; w = v.get("WRITE_SIZE", 0) / (cnt[(k, "WRITE_SIZE")] or 1)
    print(f"{k[:64]:64s} {n:8d} {f:12.1f} {w:12.1f} {(2*f+w)/1024:9.2f}")
PY
cat $R/gpurun_out/${TAG}_pmc_traffic.txt
