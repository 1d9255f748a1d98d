#!/bin/bash
# micro timings of the 16 -> 16 Winograd instances (conv3x3_tile / _persist): bash tools/wino16_micro.sh [NGAN_LIB_PATH]
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$1" ] && export NGAN_LIB_PATH=$R/$1
for args in "--B 16 --H 512 --W 512 --epi 0" "--B 16 --H 512 --W 512 --epi 1" "--B 32 --H 512 --W 512 --epi 1" "--B 32 --H 256 --W 256 --epi 1" \
            "--B 16 --H 512 --W 512 --epi 1 --res 2" "--B 32 --H 512 --W 512 --epi 1 --res 2" "--B 16 --H 256 --W 256 --epi 0 --out 1"; do
  python3 $R/tools/conv_micro.py --K 16 --N 16 $args --iters 30 | tail -1
done
