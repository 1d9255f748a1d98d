"""Per-step table from a rocprofv3 kernel_stats.csv:  python profiles/summarize.py <csv> <iterations> [min_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iters = float(sys.argv[2])
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
tot = 0.0
fam = {}
for r in rows:
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    ms = float(r["TotalDurationNs"]) / 1e6 / iters
    tot += ms
    fam[n.split("<")[0].split("(")[0]] = fam.get(n.split("<")[0].split("(")[0], 0.0) + ms
    if ms >= min_ms:
        print(f"{n[:78]:78s} calls/step={int(r['Calls']) / iters:6.1f} avg={float(r['AverageNs']) / 1e3:8.1f}us {ms:6.2f} ms/step")
print(f"TOTAL kernel time per step: {tot:.2f} ms")
print("by family:", ", ".join(f"{k}={v:.2f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:14]))
