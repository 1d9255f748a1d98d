#!/bin/bash
# micro timings of the Winograd 32-channel kernels (conv3x3_wino.hip) + PMC counters of the 32 -> 32 forward instance
R=${GRAFT_REPO_ROOT:-$(pwd)}
for args in "--B 32 --H 128 --W 128 --K 32 --N 32 --epi 1" "--B 32 --H 128 --W 128 --K 32 --N 32 --epi 0" "--B 16 --H 128 --W 128 --K 32 --N 32 --epi 1" \
            "--B 32 --H 64 --W 64 --K 32 --N 32 --epi 1" "--B 16 --H 256 --W 256 --K 32 --N 16 --epi 0" "--B 16 --H 256 --W 256 --K 16 --N 32 --epi 0" \
            "--B 32 --H 128 --W 128 --K 32 --N 16 --epi 0 --out 1"; do
  python3 $R/tools/conv_micro.py $args --iters 30 | tail -1
done
bash $R/tools/pmc_conv.sh ${1:-w32} --B 32 --H 128 --W 128 --K 32 --N 32 --epi 1 --iters 5
