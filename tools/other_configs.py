"""The bench at the shapes of BASELINE.json's other configs (C1 - C5 and two more), the three arithmetic modes in one line per shape:
    python tools/other_configs.py > gpurun_out/<tag>_other_configs.txt      (on the GPU box; record: profiles/r04_other_configs.txt)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = ["--res 16 --batch 16 --alpha 1.0", "--res 64 --batch 64 --alpha 0.5", "--res 256 --batch 32 --alpha 1.0", "--res 512 --batch 8 --alpha 1.0",
           "--res 512 --batch 16 --alpha 0.5", "--res 128 --batch 64 --alpha 1.0"]
print("python bench.py <shape flags> --steps 20 --warmup 5 --no-cpu-baseline --live-traffic 0 on one MI355X: the shapes of BASELINE.json's configs C1 - C5 "
      "(and two more) in the three arithmetic modes: exact fp32 (headline), split-bf16 and bf16 activation storage (sub-records)", flush=True)
for cfg in CONFIGS:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *cfg.split(), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--live-traffic", "0"],
                       capture_output=True, text=True, timeout=600)
    print("==", cfg, flush=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        print("   failed:", (r.stderr or r.stdout)[-400:], flush=True)
        continue
    rf = d["roofline"]
    line = (f"f32: {d['value']:.1f} images/s, {d['ms_per_step']:.3f} ms/iteration, step {d.get('step_tflops', 0):.1f} TF ({d.get('step_frac_of_fp32_mfma_peak', 0):.2f} of fp32 MFMA "
            f"peak); dominant {rf['kernel']} frac {rf['frac']:.2f} ({rf['bound']})")
    for m in ("bf16x3", "bf16"):
        s = d.get(m)
        if s:
            line += f"; {m}: {s['value']:.1f} images/s, {s['ms_per_step']:.3f} ms"
            if m == "bf16":
                line += f" ({s.get('step_frac_of_hbm_peak', 0):.2f} of HBM peak over the step; dominant {s['roofline']['kernel']} {s['roofline']['frac']:.2f})"
    print(line, flush=True)
