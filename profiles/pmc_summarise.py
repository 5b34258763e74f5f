#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes written by profiles/pmc_pass.sh.

usage: profiles/pmc_summarise.py <pmc outdir> <out.json> [kernel-substring ...]
Mean counter value per launch for every kernel whose name contains one of the substrings.
"""
import csv, glob, json, os, sys
from collections import defaultdict

def main():
    root, out = sys.argv[1], sys.argv[2]
    subs = sys.argv[3:] or ["ordered_", "fixup_kernel"]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        names = {}
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if not any(s in k for s in subs):
                    continue
                did = row.get("Dispatch_Id")
                names[did] = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                per_dispatch[did][row["Counter_Name"]] += float(row["Counter_Value"])
        for did, cs in per_dispatch.items():
            for c, v in cs.items():
                acc[names[did]][c].append(v)
    res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
    launches = {k: max(len(v) for v in cs.values()) for k, cs in acc.items()}
    json.dump({"counters_mean_per_launch": res, "launches_seen": launches}, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))

if __name__ == "__main__":
    main()
