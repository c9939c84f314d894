"""Fold rocprofv3 --pmc CSVs (separate passes) into per-kernel, per-launch means.  usage: pmc_summary.py <dir>"""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> counter -> dispatch -> value
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = re.sub(r"^void |cls::\(anonymous namespace\)::|\(.*$", "", row["Kernel_Name"])
            d = per[name][row["Counter_Name"]]
            key = (path, row["Dispatch_Id"])
            d[key] = d.get(key, 0.0) + float(row["Counter_Value"])
out = {}
for k, counters in per.items():
    if not any(s in k for s in ("place_", "order_key", "classify", "DeviceRadixSort", "radix", "Onesweep", "onesweep")):
        continue
    out[k] = {}
    for c, d in counters.items():
        vals = sorted(d.values())
        big = [v for v in vals if v >= 0.5 * vals[-1]] if vals and vals[-1] > 0 else vals  # full-size launches only (bench steps)
        out[k][c] = {"per_launch_mean": sum(big) / max(1, len(big)), "launches": len(big)}
# the dominant kernel = the placement kernel with the most fetched bytes over all its launches (the bench's timed
# steps; the single statistics launch of the untimed preamble is a different template instance)
dom = max((k for k in out if k.startswith("place_")),
          key=lambda k: out[k].get("FETCH_SIZE", {}).get("per_launch_mean", 0) * out[k].get("FETCH_SIZE", {}).get("launches", 0), default=None)
res = {
    "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes) -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline",
    "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; TCC_* in requests",
    "kernels": out,
}
if dom:
    f = out[dom].get("FETCH_SIZE", {}).get("per_launch_mean", 0.0)
    w = out[dom].get("WRITE_SIZE", {}).get("per_launch_mean", 0.0)
    res["dominant_kernel"] = dom
    res["dominant_kernel_hbm_bytes_per_launch"] = (f + w) * 1024.0
print(json.dumps(res, indent=1))
