#!/bin/bash
# kernel traces of both arithmetic modes (eager launches so every kernel is traced); run on the GPU box via gpurun
#   bash tools/measure_f32.sh <tag>
tag=${1:-r02_a}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for prec in f32 bf16x3; do
  BENCH="python3 $R/bench.py --no-cpu-baseline --no-probe --graph 0 --sub-record 0 --precision $prec"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_${prec}_stats -- $BENCH --steps 5 --warmup 2 > $R/gpurun_out/${tag}_${prec}_stats.log 2>&1) || exit 1
  cp $(ls $R/gpurun_out/${tag}_${prec}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_${prec}_kernel_stats.csv
done
