#!/usr/bin/env python3
"""Per-launch durations of the headline kernel out of a rocprofv3 --kernel-trace directory.

usage: profiles/trace_headline.py <rocprof outdir> <out.csv> [kernel-substring]
bench.py --no-extra --no-cpu-baseline launches the kernel 1 (first call) + 30 (untimed clock ramp) + warmup + steps
times; the last warmup + steps launches are the region bench.py times.
"""
import csv, glob, os, sys


def main():
    root, out = sys.argv[1], sys.argv[2]
    sub = sys.argv[3] if len(sys.argv) > 3 else "ordered_lean_kernel<1, 8"
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if sub in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    n = len(rows)
    region = n - 31
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["launch", "kernel", "duration_us", "note"])
        for i, (s, e, k) in enumerate(rows):
            name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(" ", "")
            note = "first call" if i == 0 else "untimed clock ramp" if i <= 30 else \
                f"profiled region of bench.py ({region} launches: warm-up + timed steps)"
            w.writerow([i, name, f"{(e - s) / 1e3:.1f}", note])
    tail = [(e - s) / 1e3 for s, e, _ in rows[31:]]
    print(f"{n} launches; profiled region: {len(tail)} launches, mean {sum(tail) / max(len(tail), 1):.1f} us; "
          f"all: mean {sum((e - s) / 1e3 for s, e, _ in rows) / max(n, 1):.1f} us")


if __name__ == "__main__":
    main()
