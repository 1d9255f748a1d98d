export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_aw_graph -- python3 $R/bench.py --no-cpu-baseline --no-probe --sub-record 0 --steps 10 --warmup 2 > $R/gpurun_out/r02_aw_graph.log 2>&1
tail -1 $R/gpurun_out/r02_aw_graph.log | cut -c1-200
