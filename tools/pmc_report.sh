#!/bin/bash
# Per-kernel evidence table for the roofline pass (python bench.py --roofline-only): duration (kernel trace), HBM bytes
# (FETCH_SIZE, WRITE_SIZE: separate --pmc passes) -> GB/s vs the 8 TB/s roof, and MFMA-busy for the MFMA kernels.
# Every --pmc pass is its own rocprofv3 run without any trace domain.   usage: tools/pmc_report.sh TAG
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
CMD="python bench.py --roofline-only --steps 2 --warmup 1 $BENCH_ARGS"   # BENCH_ARGS="--dtype bf16": the configs[2] pass
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rep_${TAG}_trace -o t -- $CMD > /tmp/rep_${TAG}_trace.log 2>&1)
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $grp --output-format csv -d /tmp/rep_${TAG}_pmc$i -- $CMD > /tmp/rep_${TAG}_pmc$i.log 2>&1) || tail -3 /tmp/rep_${TAG}_pmc$i.log
done
cd $R && python - "$TAG" > $R/gpurun_out/${TAG}_pmc_report.txt <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
dur = {}
for f in glob.glob(f"/tmp/rep_{tag}_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for i in (1, 2, 3):
    for f in glob.glob(f"/tmp/rep_{tag}_pmc{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
def per(k, c):
    n = cnt[(k, c)]
    return acc[k][c] / n if n else float("nan")
print("# command: rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES}")
import os
print("#          -- python bench.py --roofline-only --steps 2 --warmup 1 " + os.environ.get("BENCH_ARGS", "") + "     (4 separate runs; MI355X, gfx950)")
print("# hbm_MB/launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB / 1024  (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half of wide coalesced reads;")
print("#                 uncalibrated for narrow accesses -> read GB/s as an upper estimate); GB/s = hbm bytes / average kernel duration; roof 8000 GB/s (6300 achievable)")
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs): share of SIMD cycles with the fp32 MFMA pipe busy")
print(f"{'kernel':80s} {'calls':>6s} {'avg_us':>8s} {'tot_ms':>8s} {'hbm_MB':>8s} {'GB/s':>7s} {'of8TB/s':>7s} {'mfma_busy':>9s}")
rows = sorted(dur.items(), key=lambda kv: -kv[1][2])
for k, (calls, avg, tot) in rows[:40]:
    kk = k
    f, w = per(kk, "FETCH_SIZE"), per(kk, "WRITE_SIZE")
    mb = (2 * f + w) / 1024 if f == f and w == w else float("nan")
    gbs = mb * 1e6 / (avg * 1e-6) / 1e9 if mb == mb else float("nan")
    mf, gui = per(kk, "SQ_VALU_MFMA_BUSY_CYCLES"), per(kk, "GRBM_GUI_ACTIVE")
    busy = mf / (gui * 128) if (mf == mf and gui == gui and gui > 0 and mf > 0) else float("nan")
    print(f"{k[:80]:80s} {calls:6d} {avg:8.1f} {tot:8.2f} {mb:8.2f} {gbs:7.0f} {gbs/8000 if gbs==gbs else float('nan'):7.3f} {busy:9.3f}")
PY
head -40 $R/gpurun_out/${TAG}_pmc_report.txt | cut -c1-150
