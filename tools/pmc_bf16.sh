#!/bin/bash
# PMC passes (counters only, one rocprofv3 run per group) over one shape of tools/bf16_micro.py; run on the GPU box via gpurun:
#   bash tools/pmc_bf16.sh <tag> "<substring of the shape's name>"
tag=$1; only=$2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/tools/bf16_micro.py --iters 5 --only "$only" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1) || exit 1
done
python3 - "$R" "$tag" <<'PY'
import csv, glob, sys
R, tag = sys.argv[1], sys.argv[2]
tot = {}
for f in glob.glob(f"{R}/gpurun_out/pmc_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if ("conv3x3" not in r["Kernel_Name"] and "wgrad_" not in r["Kernel_Name"]) or "pack" in r["Kernel_Name"] or "reduce" in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][-60:], r["Counter_Name"])
        d = tot.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += float(r["Counter_Value"])
for (kn, cn), (n, v) in sorted(tot.items()):
    print(f"{kn:62s} {cn:28s} launches={n:3d} per-launch={v / n:16.1f}")
PY
