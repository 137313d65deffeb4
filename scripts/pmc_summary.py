"""Summarise two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE -- collected separately, with --kernel-trace only,
as MI355X_MICROARCH.md prescribes) into profiles/rNN_scan_pmc.json.

  python scripts/pmc_summary.py <fetch_dir> <write_dir> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in units of 1024 B; on gfx950 FETCH_SIZE under-counts wide (16 B / lane)
coalesced reads by 2x (guide, HBM section), so the corrected figure is reported beside the raw one.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        a = acc[name]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith(("lh::", "void lh::")):
            continue
        kernels[k] = {"FETCH_SIZE_avg_per_launch": fe.get(k, (0, 0))[0], "launches_FETCH_SIZE": fe.get(k, (0, 0))[1],
                      "WRITE_SIZE_avg_per_launch": wr.get(k, (0, 0))[0], "launches_WRITE_SIZE": wr.get(k, (0, 0))[1]}
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 5 "
                     "--no-cpu-baseline, MI355X",
           "units": "FETCH_SIZE/WRITE_SIZE in units of 1024 B; gfx950 correction: FETCH_SIZE x2 for wide coalesced reads "
                    "(MI355X_MICROARCH.md, HBM section)",
           "kernels": kernels}
    # the two launches of the partition-major scan: RPL = 0 (bound pass) and RPL = 2 (main pass)
    for k, v in kernels.items():
        if "ivfpq_scan_pm_kernel" in k:
            tag = "scan_bound_pass_kernel" if ", 0, " in k.split("<")[1] and k.split("<")[1].split(",")[3].strip() == "0" else "scan_main_pass_kernel"
            raw = v["FETCH_SIZE_avg_per_launch"] * 1024
            res[tag] = {"name": k, "fetch_bytes_raw": raw, "fetch_bytes_corrected_x2": 2 * raw,
                        "write_bytes": v["WRITE_SIZE_avg_per_launch"] * 1024}
    if "scan_main_pass_kernel" in res:
        m = res["scan_main_pass_kernel"]
        res["hbm_bytes_per_launch"] = m["fetch_bytes_corrected_x2"] + m["write_bytes"]
    json.dump(res, open(out, "w"), indent=1)
    print({k: v for k, v in res.items() if k != "kernels"})


if __name__ == "__main__":
    main()
