"""launch census of a rocprofv3 --kernel-trace --stats run: kernels ordered by CALL COUNT per step, and the time of the
"tiny" ones (avg < 10 us).  usage: python tools/kernel_census.py <rocprof dir> <steps>"""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0])))
rows = [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in rows]
tot_calls, tot_ms = sum(r[1] for r in rows), sum(r[2] for r in rows)
tiny = [(n, c, ms) for n, c, ms in rows if 1e3 * ms / c < 10.0]
print(f"total {tot_ms/steps:.2f} ms/step over {tot_calls/steps:.0f} launches/step; "
      f"tiny (<10us avg): {sum(r[2] for r in tiny)/steps:.2f} ms/step over {sum(r[1] for r in tiny)/steps:.0f} launches/step")
for n, c, ms in sorted(rows, key=lambda r: -r[1])[:45]:
    print(f"{c/steps:8.1f}/step {ms/steps:7.3f} ms/step avg {1e3*ms/c:6.1f} us  {n[:110]}")
