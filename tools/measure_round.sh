#!/bin/bash
# The measurement set behind DESIGN.md / profiles/ (run on the GPU box through gpurun from the repo root):
#   bash tools/measure_round.sh <tag>          e.g. r02
# for each arithmetic mode (f32 = the headline, bf16x3 and bf16 = the labelled sub-records):
#   1. rocprofv3 --kernel-trace --stats over the bench command (eager launches, so every kernel is traced)
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over the same command, counters only
#   3. tools/traffic_summary.py -> gpurun_out/<tag>_traffic_<mode>.json (bytes per launch and kernel, gfx950 correction applied)
# then 4. the default bench line (fp32 headline + split-bf16 sub-record, HIP-graph replay, probe, CPU baseline)
tag=${1:-r04}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for prec in f32 bf16x3 bf16; do
  BENCH="python3 $R/bench.py --no-cpu-baseline --no-probe --graph 0 --sub-record 0 --precision $prec"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_${prec}_stats -- $BENCH --steps 5 --warmup 2 > $R/gpurun_out/${tag}_${prec}_stats.log 2>&1) || exit 1
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_${prec}_fetch -- $BENCH --steps 2 --warmup 1 > $R/gpurun_out/${tag}_${prec}_fetch.log 2>&1) || exit 1
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_${prec}_write -- $BENCH --steps 2 --warmup 1 > $R/gpurun_out/${tag}_${prec}_write.log 2>&1) || exit 1
  python3 $R/tools/traffic_summary.py $R/gpurun_out/${tag}_${prec}_fetch $R/gpurun_out/${tag}_${prec}_write $R/gpurun_out/${tag}_traffic_${prec}.json || exit 1
  cp $(ls $R/gpurun_out/${tag}_${prec}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_kernel_stats_${prec}.csv
done
python3 $R/bench.py > $R/gpurun_out/${tag}_bench.log 2>&1
tail -1 $R/gpurun_out/${tag}_bench.log | cut -c1-600
