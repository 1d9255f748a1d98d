"""Per-iteration summary of a rocprofv3 kernel_stats.csv of tools/measure_round.sh / tools/stats_mode.sh (7 eager iterations traced:
--steps 5 --warmup 2):   python tools/stats_summary.py gpurun_out/<tag>_kernel_stats_<mode>.csv <mode> [tag] > profiles/<tag>_kernel_stats_<mode>_summary.txt
Also prints the split conv family / everything else that DESIGN.md section 5 quotes."""
import csv
import sys

CONV = ("conv3x3", "wgrad_", "first_block", "pack_weights")


def main():
    path, mode = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r04"
    rows = list(csv.DictReader(open(path)))
    n_it = 7.0
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    conv = sum(float(r["TotalDurationNs"]) for r in rows if any(c in r["Name"] for c in CONV) and "reduce" not in r["Name"])
    print(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-probe --graph 0 --sub-record 0 --precision {mode} --steps 5 --warmup 2   (tools/measure_round.sh {tag})")
    print(f"total kernel time {tot / n_it / 1e6:.3f} ms per iteration ({int(n_it)} iterations traced, eager launches); "
          f"3x3-conv family (forward, input and weight gradients, first block, weight packing) {conv / n_it / 1e6:.3f} ms, everything else {(tot - conv) / n_it / 1e6:.3f} ms")
    for r in rows[:60]:
        print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]) / n_it:6.1f}/it  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {float(r["TotalDurationNs"]) / n_it / 1e3:8.1f} us/it  {float(r["Percentage"]):5.2f} %')


if __name__ == "__main__":
    main()
