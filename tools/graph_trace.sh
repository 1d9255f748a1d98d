#!/bin/bash
# kernel trace of GRAPH-REPLAYED iterations: bash tools/graph_trace.sh <tag>   (on the GPU box through gpurun)
# prints, for the last replayed iteration: wall time from its first kernel's start to its last kernel's end, the sum of the kernel
# durations inside it, the idle time between kernels and the number of launches
tag=${1:-r03}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_graph_trace -- python3 $R/bench.py --no-cpu-baseline --no-probe --sub-record 0 --live-traffic 0 --steps 10 --warmup 2 > $R/gpurun_out/${tag}_graph_trace.log 2>&1) || exit 1
python3 - $R/gpurun_out/${tag}_graph_trace <<'PY'
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
# an iteration ends with the generator's flat Adam launch followed by its re-pack launch: cut between consecutive pack_many after adam pairs
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
g_adam = idx[1::2]                       # every second adam_kernel is the generator's
for k in (-3, -2):
    a, b = g_adam[k] + 2, g_adam[k + 1] + 2          # (+2: adam_kernel, then pack_many_kernel close the iteration)
    it = rows[a:b]
    t0, t1 = int(it[0]["Start_Timestamp"]), int(it[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it)
    gaps = [int(it[i + 1]["Start_Timestamp"]) - int(it[i]["End_Timestamp"]) for i in range(len(it) - 1)]
    gaps_pos = [g for g in gaps if g > 0]
    print(f"replayed iteration: {len(it)} launches, wall {(t1 - t0) / 1e6:.3f} ms, kernel durations summed {busy / 1e6:.3f} ms, "
          f"idle between kernels {sum(gaps_pos) / 1e6:.3f} ms (median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, "
          f"{sum(1 for g in gaps if g > 2000)} gaps above 2 us)")
PY
