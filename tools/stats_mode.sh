#!/bin/bash
# kernel-trace statistics of eager iterations in one arithmetic mode: bash tools/stats_mode.sh <tag> <f32|bf16x3|bf16> [extra bench args]
# (on the GPU box through gpurun; the summary it prints is what gets committed under profiles/)
tag=${1:-r04}
prec=${2:-bf16}
shift 2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH="python3 $R/bench.py --no-cpu-baseline --no-probe --graph 0 --sub-record 0 --precision $prec $*"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_${prec}_stats -- $BENCH --steps 5 --warmup 2 > $R/gpurun_out/${tag}_${prec}_stats.log 2>&1) || exit 1
cp $(ls $R/gpurun_out/${tag}_${prec}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_kernel_stats_${prec}.csv
python3 - $R/gpurun_out/${tag}_kernel_stats_${prec}.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 7e6:.3f} ms per iteration (7 iterations traced)")
for r in rows[:50]:
    print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]) / 7:6.1f}/it  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {float(r["TotalDurationNs"]) / 7e3:8.1f} us/it  {float(r["Percentage"]):5.2f} %')
PY
