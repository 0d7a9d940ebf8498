"""Aggregate kernel-trace launches of one kernel family by (grid, lds) to find the expensive shapes."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
rows = list(csv.DictReader(open(f)))
cols = rows[0].keys()
for r in rows:
    if pat not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"][:44], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r.get("LDS_Block_Size") or r.get("LDS_Block_Size_v"))
    a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
print("columns:", [c for c in cols if "Size" in c or "LDS" in c])
print(f"total {tot/1e3:.2f} ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[0]:44s} grid=({k[1]},{k[2]},{k[3]}) lds={k[4]:>7s} n={a[0]:5d} total={a[1]/1e3:8.2f} ms avg={a[1]/a[0]:8.1f} us {100*a[1]/tot:5.1f}%")
