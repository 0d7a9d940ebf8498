"""Summarise a rocprofv3 --kernel-trace --stats run (csv output) into a small text table."""
import csv, glob, sys
d = sys.argv[1]
fs = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {fs[0]}\n# total kernel time {tot/1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches")
print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
