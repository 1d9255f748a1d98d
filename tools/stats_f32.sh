#!/bin/bash
# kernel-trace statistics of eager fp32 iterations: bash tools/stats_f32.sh <tag>   (on the GPU box through gpurun)
tag=${1:-r03}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH="python3 $R/bench.py --no-cpu-baseline --no-probe --graph 0 --sub-record 0 --precision f32"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_f32_stats -- $BENCH --steps 5 --warmup 2 > $R/gpurun_out/${tag}_f32_stats.log 2>&1) || exit 1
cp $(ls $R/gpurun_out/${tag}_f32_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_kernel_stats_f32.csv
python3 - $R/gpurun_out/${tag}_kernel_stats_f32.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 7e6:.3f} ms per iteration (7 iterations traced)")
for r in rows[:45]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]) / 7:6.1f}/it  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {float(r["TotalDurationNs"]) / 7e3:8.1f} us/it  {float(r["Percentage"]):5.2f} %')
PY
