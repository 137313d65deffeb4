"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, the per-launch average of every counter collected.

  python scripts/pmc_sq_summary.py <rocprof_out_dir> <out.json> [kernel-name-substring]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1:3]
    sub = sys.argv[3] if len(sys.argv) > 3 else "lh::"
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if sub not in name:
                continue
            a = acc[name.split("(")[0]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    res = {k: {c: {"avg_per_launch": v[0] / v[1], "launches": v[1]} for c, v in cs.items()} for k, cs in acc.items()}
    json.dump(res, open(out, "w"), indent=1)
    for k, cs in res.items():
        print(k[:110], {c: round(v["avg_per_launch"]) for c, v in cs.items()})


if __name__ == "__main__":
    main()
