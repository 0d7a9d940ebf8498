"""profiles/<tag>_pmc_report.txt (tools/pmc_report.sh) -> profiles/latest_traffic.json, the file bench.py reads
roofline.traffic / mfma_busy_pmc from (PMC counters cannot be read from inside the process).
usage: python tools/make_traffic_json.py profiles/<tag>_pmc_report.txt [profiles/<tag>_bf16_pmc_report.txt ...]"""
import json, re, sys


def parse_report(path, into, tag_source=False):
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel") or not line.strip():
            continue
        parts = line.rstrip().rsplit(None, 7)  # name (may contain spaces) + 7 numeric columns
        if len(parts) < 8:
            continue
        name, (calls, avg, tot, mb, gbs, frac, busy) = parts[0].strip(), parts[1:]
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name).strip()  # drop the argument list (possibly cut by the column width)
        if "conv_" not in name or mb == "nan":
            continue
        into[name] = {"launches": int(calls), "avg_us": float(avg), "hbm_bytes_per_launch": int(float(mb) * 1e6),
                      "hbm_GBps": float(gbs), "mfma_busy": None if busy == "nan" else float(busy)}
        if tag_source:
            into[name]["source"] = path


src = sys.argv[1]
kern = {}
parse_report(src, kern)
for extra in sys.argv[2:]:  # further reports in the same format (the bf16 pass: BENCH_ARGS="--dtype bf16" tools/pmc_report.sh TAG)
    parse_report(extra, kern, tag_source=True)
json.dump({"source": src, "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
           "(separate passes) -- python bench.py --roofline-only --steps 2 --warmup 1 [--dtype bf16]",
           "correction": "(2 x FETCH_SIZE + WRITE_SIZE) KiB -> bytes, MI355X_MICROARCH.md HBM section", "kernels": kern},
          open("profiles/latest_traffic.json", "w"), indent=1)
print(len(kern), "kernels")
