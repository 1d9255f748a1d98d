"""One training iteration as a time-ordered kernel list, from a `rocprofv3 --kernel-trace` run of eager iterations
(tools/measure_round.sh writes gpurun_out/<tag>_<mode>_stats/*/*_kernel_trace.csv).  The last complete iteration is cut out between
Adam launches: D step (ends with the critic's adam_kernel) + G step (ends with the generator's).

    python tools/iteration_trace.py gpurun_out/r02_f32_stats > profiles/r02_iteration_trace_f32.txt
"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:86]


def main(run_dir):
    path = max(glob.glob(run_dir + "/*/*_kernel_trace.csv"), key=os.path.getmtime)       # the newest run in the directory
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    adam = [i for i, n in enumerate(names) if n.startswith("adam_kernel")]
    first, last = adam[-3] + 1, adam[-1] + 1             # after the previous G update .. this iteration's G update
    t0 = int(rows[first]["Start_Timestamp"])
    busy = 0.0
    print(f"# {path}: launches {first}..{last - 1} (one iteration: critic step, then generator step); columns: start us, duration us, kernel, threads in x")
    for i in range(first, last):
        r = rows[i]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        busy += d
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:7.1f}  {names[i]}  {r.get('Grid_Size_X', '')}")
    print(f"# {last - first} launches, {busy / 1e3:.2f} ms of kernel time (eager launches under the profiler: the gaps are host time and do not exist in graph replay)")


if __name__ == "__main__":
    main(sys.argv[1])
