"""GPU busy vs wall from a rocprofv3 kernel trace (csv): overall and for the trailing 60% of the run."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
n = len(rows); tail = rows[int(n * 0.4):]
for name, rs in (("all", rows), ("tail60%", tail)):
    span = rs[-1][1] - rs[0][0]; busy = sum(e - s for s, e, _ in rs)
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    big = sum(g for g in gaps if g > 20000)
    print(f"{name}: launches {len(rs)} span {span/1e6:.1f} ms busy {busy/1e6:.1f} ms ({100*busy/span:.1f}%) gaps>20us total {big/1e6:.1f} ms; mean gap {sum(max(g,0) for g in gaps)/len(gaps)/1e3:.2f} us")
