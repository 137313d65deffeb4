"""HBM traffic per kernel launch from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE collected separately, with
--kernel-trace only, as MI355X_MICROARCH.md prescribes: FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2).

  python scripts/pmc_tcc_summary.py <fetch_dir> <write_dir> <out.json> ["command that was profiled"]

Units / corrections (guide, HBM section): both counters are in units of 1024 B; on gfx950 FETCH_SIZE reports exactly half of
the bytes of wide (16 B per lane) coalesced streaming reads, so the corrected figure (x2) is given beside the raw one.  For
kernels whose loads are narrower the truth lies between the two; WRITE_SIZE is uncalibrated and reported raw.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"].split("(")[0]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fd, wd, out = sys.argv[1:4]
    cmd = sys.argv[4] if len(sys.argv) > 4 else ""
    fe, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if "lh::" not in k:
            continue
        f_raw = fe.get(k, (0.0, 0))[0] * 1024
        w_raw = wr.get(k, (0.0, 0))[0] * 1024
        kernels[k] = {"launches": fe.get(k, (0, 0))[1] or wr.get(k, (0, 0))[1],
                      "fetch_bytes_per_launch_raw": f_raw, "fetch_bytes_per_launch_x2": 2 * f_raw,
                      "write_bytes_per_launch": w_raw, "hbm_bytes_per_launch": 2 * f_raw + w_raw}
    res = {"source": f"rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) -- {cmd}",
           "units": "bytes per launch (averages over the launches of the run); hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE "
                    "(gfx950 FETCH_SIZE correction for wide coalesced reads, MI355X_MICROARCH.md HBM section)",
           "kernels": kernels}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:25]:
        print(f"{k[:100]:100s} {v['launches']:5d} launches  fetch x2 {v['fetch_bytes_per_launch_x2'] / 1e6:10.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
