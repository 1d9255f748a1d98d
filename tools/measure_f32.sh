#!/bin/bash
# f32-mode kernel trace (eager launches so every kernel is traced) + default-mode bench lines; run on the GPU box via gpurun
tag=${1:-r02_a}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH="python3 $R/bench.py --no-cpu-baseline --no-probe --graph 0 --precision f32"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_f32_stats -- $BENCH --steps 5 --warmup 2 > $R/gpurun_out/${tag}_f32_stats.log 2>&1) || exit 1
cp $(ls $R/gpurun_out/${tag}_f32_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_f32_kernel_stats.csv
python3 $R/bench.py --precision f32 --no-cpu-baseline > $R/gpurun_out/${tag}_f32_bench.log 2>&1
tail -1 $R/gpurun_out/${tag}_f32_bench.log
