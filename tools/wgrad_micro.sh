#!/bin/bash
# micro timings of the fp32 weight-gradient kernels: bash tools/wgrad_micro.sh [NGAN_LIB_PATH]
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$1" ] && export NGAN_LIB_PATH=$R/$1
for args in "--B 32 --H 128 --W 128 --K 32 --N 32" "--B 16 --H 128 --W 128 --K 32 --N 32" "--B 32 --H 64 --W 64 --K 32 --N 32" "--B 16 --H 128 --W 128 --K 32 --N 32 --res 2" \
            "--B 16 --H 512 --W 512 --K 16 --N 16" "--B 32 --H 256 --W 256 --K 16 --N 16" "--B 16 --H 256 --W 256 --K 32 --N 16" "--B 32 --H 16 --W 16 --K 128 --N 128" "--B 32 --H 32 --W 32 --K 64 --N 64"; do
  python3 $R/tools/conv_micro.py --op wgrad $args --iters 30 | tail -1
done
