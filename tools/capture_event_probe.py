"""RECORD ONLY -- do not run on the GPU pool again.  Three of its four cases end in the process-group watchdog's std::terminate
by design (an RCCL group alive, an active capture); the cause they established is fixed by construction
(PGGANTrainer._on_comm_stream + thread_local capture mode), the surviving case is covered by
tests/test_gpu_dist.py::test_segmented_capture_with_a_live_rccl_group_matches_eager, and the output of the one run that was needed is
profiles/r02_capture_event_probe.txt.

Root-cause probe for the abort on record in gpurun_out/fd2.log (round 1):

    Process group watchdog thread terminated with exception: HIP error: operation not permitted on an event last recorded in a
    capturing stream (hipErrorCapturedEvent)   raised from WorkNCCL::finishedGPUExecutionInternal -> ncclEndEvent_->query()

Hypothesis read off that stack: (1) a synchronous c10d collective (async_op=False) runs on the CALLER's current stream and records
its completion event there; (2) the process group's watchdog thread keeps polling that event with hipEventQuery until one poll
finds it complete (polling period ~100 ms); (3) HIP refuses hipEventQuery on an event whose last-recorded stream is capturing NOW --
even if the record itself happened before the capture began; (4) during a captured backward pass, autograd's AccumulateGrad
stream synchronisation forks the stream the parameters were first used on into the capture.  So a collective issued on that stream
just before a capture is a time bomb with a ~100 ms fuse.

Each case runs in a child process (an abort is the expected outcome of some), exactly once:
    work_stream/global        collective on the stream that later joins the capture, default capture mode      -> expect ABORT
    work_stream/thread_local  the same in thread_local capture mode (does the mode matter for this rule?)
    comm_stream/global        collective on a dedicated stream that never takes part in a capture: the event rule no longer
                              applies, but in GLOBAL capture mode HIP refuses hipEventQuery from ANY thread while a capture is
                              active (hipErrorStreamCaptureUnsupported)                                        -> expect ABORT
    comm_stream/thread_local  dedicated stream + thread_local capture mode (the fix: both rules satisfied)    -> expect survive
Run on the GPU box:  python tools/capture_event_probe.py
"""
import os
import subprocess
import sys
import time


def child(where, mode):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    x = torch.ones(1 << 20, device=dev)
    dist.all_reduce(x)                      # communicator set-up
    torch.cuda.synchronize()
    time.sleep(0.5)                         # the watchdog reaps the set-up work
    s_work, s_comm, cap = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s_comm if where == "comm_stream" else s_work):
        dist.all_reduce(x)                  # sync op: completion event recorded on the current stream
    with torch.cuda.stream(cap):
        g.capture_begin(capture_error_mode=mode)
        y = x * 2
        ev = torch.cuda.Event()
        ev.record(cap)
        s_work.wait_event(ev)               # s_work forks into the capture (what AccumulateGrad's stream sync does)
        with torch.cuda.stream(s_work):
            z = y + 1
        time.sleep(0.6)                     # several watchdog polls while s_work is part of an active capture
        cap.wait_stream(s_work)
        g.capture_end()
    g.replay()
    torch.cuda.synchronize()
    assert float(z[0]) == 3.0
    dist.destroy_process_group()
    print("SURVIVED", where, mode, flush=True)


if __name__ == "__main__":
    import os as _os, sys as _sys
    if _os.environ.get("NGAN_ALLOW_ABORTING_PROBE") != "1":
        _sys.exit("record-only tool (provokes process aborts on the GPU box); see the docstring")
    if len(sys.argv) == 3:
        child(sys.argv[1], sys.argv[2])
        sys.exit(0)
    for where, mode in (("work_stream", "global"), ("work_stream", "thread_local"), ("comm_stream", "global"), ("comm_stream", "thread_local")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), where, mode], capture_output=True, text=True, timeout=300)
        tail = [ln for ln in (r.stdout + r.stderr).splitlines() if "hipError" in ln or "SURVIVED" in ln or "terminate" in ln]
        print(f"case {where}/{mode}: rc={r.returncode} " + ("SURVIVED" if "SURVIVED" in r.stdout else "ABORTED") + " | " + " | ".join(t[:160] for t in tail[:3]),
              flush=True)
