"""HBM bytes per launch and kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over the bench command.
    python tools/traffic_summary.py <fetch_dir> <write_dir> <out.json>
Corrections (MI355X_MICROARCH.md, HBM / rocprofv3): both counters are reported in KiB; on gfx950 FETCH_SIZE counts a wide
(16 B per lane) streaming read at half its bytes (128-B requests tallied as 64 B), so it is doubled; WRITE_SIZE is exact."""
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            e = out.setdefault(name, {})
            e[r["Dispatch_Id"]] = e.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])     # summed over the 8 XCDs
    return {k: (len(v), sum(v.values())) for k, v in out.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
kernels = {}
for name in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(name, (0, 0.0))
    nw, w = write.get(name, (0, 0.0))
    n = max(nf, nw)
    rd = 2.0 * f * 1024.0 / max(nf, 1)
    wr = w * 1024.0 / max(nw, 1)
    kernels[name] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 "
                      "--no-cpu-baseline --no-probe --graph 0 --sub-record 0 --precision <mode> (tools/measure_round.sh)",
           "correction": "FETCH_SIZE x2 (gfx950 reports half of a wide streaming read), KiB -> bytes", "kernels": kernels},
          open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(kernels), "kernels")
