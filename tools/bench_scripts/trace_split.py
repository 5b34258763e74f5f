"""Per-kernel durations out of a rocprofv3 --kernel-trace csv, grouped by the n-th block of `per` launches of each kernel (one block per
workload of the traced script).   usage: trace_split.py kernel_trace.csv name-substring per label0,label1,..."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub, per, labels = sys.argv[2], int(sys.argv[3]), sys.argv[4].split(",")
out, idx = collections.OrderedDict(), collections.Counter()
for r in rows:
    if sub not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("dp::", "").replace("void ", "").split("(")[0]
    i = idx[k]; idx[k] += 1
    out.setdefault((labels[min(len(labels) - 1, i // per)], k), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (lab, k), v in out.items():
    v = sorted(v)
    print(f"{lab:12s} {k:40s} n={len(v):3d} min {v[0]:8.1f} us  median {v[len(v) // 2]:8.1f} us")
