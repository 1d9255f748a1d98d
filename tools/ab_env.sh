#!/bin/bash
# usage: tools/ab_env.sh "VAR=a VAR2=b" "VAR=c" ...   -- one bench run per environment string, prints img/s and ms/step
# Python-level switches (honoured with NGAN_DIAG=1, which this script sets): NGAN_FIRST_BLOCK=0, NGAN_FIRST_ORDER_FUSION=0, NGAN_POOL_FIRST=0, NGAN_POOL_OUT=0, NGAN_LIB_PATH=<other build>.
# Kernel-dispatch switches exist in the DIAGNOSTIC library only (make -C neuron-gan_amd/csrc diag; NGAN_LIB_PATH=build/diag/libngan_hip_diag.so):
#   NGAN_UP2_FOLDED=0, NGAN_TILE_KERNEL=0, NGAN_MID_F32=0, NGAN_WINOGRAD=0, NGAN_WINOGRAD32=0, NGAN_WINOGRAD_UP2=0, NGAN_WINOGRAD_WGRAD=0,
#   NGAN_WGRAD_SLABS=<n>, NGAN_PERSIST_WG_PER_CU=<n>
for envs in "$@"; do
  out=$(env NGAN_DIAG=1 $envs timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --live-traffic 0 2>/dev/null | tail -1)
  echo "$envs :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],4), d["roofline"]["kernel"], round(d["roofline"]["frac"],3)); [print("    ",k,v["avg_us"],v["gbs"],v["launches_per_step"]) for k,v in d["conv_family"]["instances"].items() if "persist" in k or "up2f" in k]')"
done
