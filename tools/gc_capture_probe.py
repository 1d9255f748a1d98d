"""Which finalizer must not run inside a global-mode stream capture?  (round-3 record: gpurun_out/r03_a_tests.log, `Fatal Python
error: Aborted` under "Garbage-collecting" on the autograd thread during the third capture() of a process; the fix in
PGGANTrainer.capture -- collect first, keep the cyclic collector off until the capture ends -- did not name the object.)

Each case runs ONCE in a child process: a small graph is captured and kept, then a second capture (global error mode, the mode of a
one-GPU run) is opened and, while it is active, one kind of object left over from earlier work is destroyed on the capturing thread --
what a cyclic collection that happens to start there would do:
    graph      the earlier torch.cuda.CUDAGraph object (-> hipGraphExecDestroy / hipGraphDestroy, and its private pool is released)
    event      a torch.cuda.Event that was recorded earlier (-> hipEventDestroy)
    pool       a tensor that lives in the earlier graph's private memory pool
    twostream  an ordinary tensor that was used on a second stream (record_stream: the allocator wants an event for it on free)
    none       nothing (control)
The child prints SURVIVED if the capture completes and replays; an abort / HIP error is reported with its first error line.
Run on the GPU box:  python tools/gc_capture_probe.py          (record: profiles/r04_gc_capture_probe.txt)
"""
import os
import subprocess
import sys


def child(what):
    import gc
    import torch
    gc.disable()
    dev = torch.device("cuda:0")
    x = torch.ones(1 << 16, device=dev)
    side = torch.cuda.Stream()
    old = torch.cuda.CUDAGraph()
    with torch.cuda.graph(old):
        held = x * 3                      # lives in old's private pool
    old.replay()
    ev = torch.cuda.Event()
    ev.record()
    two = torch.empty(1 << 16, device=dev)
    with torch.cuda.stream(side):
        two.add_(1)
    two.record_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        g.capture_begin(capture_error_mode="global")
        y = x * 2
        if what == "graph":
            del held
            del old
        elif what == "event":
            del ev
        elif what == "pool":
            del held
        elif what == "twostream":
            del two
        z = y + 1
        g.capture_end()
    g.replay()
    torch.cuda.synchronize()
    assert float(z[0]) == 3.0
    print("SURVIVED", what, flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 2:
        child(sys.argv[1])
        sys.exit(0)
    for what in ("none", "event", "pool", "twostream", "graph"):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), what], capture_output=True, text=True, timeout=180)
            text = r.stdout + r.stderr
            tail = [ln.strip() for ln in text.splitlines() if "hipError" in ln or "HIP error" in ln or "terminate" in ln or "Aborted" in ln or "Error" in ln]
            print(f"case {what}: rc={r.returncode} " + ("SURVIVED" if "SURVIVED" in r.stdout else "FAILED") + " | " + " | ".join(t[:200] for t in tail[:3]), flush=True)
        except subprocess.TimeoutExpired:
            print(f"case {what}: TIMEOUT", flush=True)
