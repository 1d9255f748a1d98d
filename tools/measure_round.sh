#!/bin/bash
# The measurement set behind DESIGN.md / profiles/ (run on the GPU box through gpurun from the repo root):
#   bash tools/measure_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats over the bench command (eager launches, so every kernel is traced)
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over the same command, counters only
#   3. tools/traffic_summary.py -> gpurun_out/<tag>_traffic.json (bytes per launch and kernel, gfx950 correction applied)
#   4. the default bench line (HIP-graph replay, probe, CPU baseline)
tag=${1:-r01_d}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH="python3 $R/bench.py --no-cpu-baseline --no-probe --graph 0"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- $BENCH --steps 5 --warmup 2 > $R/gpurun_out/${tag}_stats.log 2>&1) || exit 1
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -- $BENCH --steps 2 --warmup 1 > $R/gpurun_out/${tag}_fetch.log 2>&1) || exit 1
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -- $BENCH --steps 2 --warmup 1 > $R/gpurun_out/${tag}_write.log 2>&1) || exit 1
python3 $R/tools/traffic_summary.py $R/gpurun_out/${tag}_fetch $R/gpurun_out/${tag}_write $R/gpurun_out/${tag}_traffic.json || exit 1
cp $(ls $R/gpurun_out/${tag}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
python3 $R/bench.py > $R/gpurun_out/${tag}_bench.log 2>&1
tail -1 $R/gpurun_out/${tag}_bench.log
