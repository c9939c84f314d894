"""Fold rocprofv3 --pmc CSVs (separate passes) into per-kernel, per-launch means.
usage: pmc_summary.py <dir> [--command "<the profiled command, as run>"]
The counter names come from the CSVs themselves; the dominant kernel is the placement kernel with the most FETCH_SIZE
over all its launches when that counter was collected, else the one with the largest total of a cycle / instruction counter (the bench's timed steps:
the single statistics launch of the untimed preamble is a different template instance)."""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
command = sys.argv[sys.argv.index("--command") + 1] if "--command" in sys.argv else None
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
place = [k for k in out if k.startswith("place_")]
have_fetch = any("FETCH_SIZE" in out[k] for k in place)
if have_fetch:
    dom = max(place, key=lambda k: out[k].get("FETCH_SIZE", {}).get("per_launch_mean", 0) * out[k].get("FETCH_SIZE", {}).get("launches", 0), default=None)
else:  # no byte counter in these passes: the instance the timed steps launched most often
    def weight(k):  # total of the first counter that says how much the kernel ran
        for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "TCC_REQ_sum", "TCP_TCC_READ_REQ_sum"):
            if c in out[k]:
                return out[k][c]["per_launch_mean"] * out[k][c]["launches"]
        return max((c["launches"] for c in out[k].values()), default=0)
    dom = max(place, key=weight, default=None)
counters = sorted({c for k in out for c in out[k]})
res = {
    "command": command or "(not recorded)",
    "counters": counters,
    "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; TCC_* / TCP_* in requests; SQ_* in instructions / cycles",
    "kernels": out,
}
if dom:
    f = out[dom].get("FETCH_SIZE", {}).get("per_launch_mean", 0.0)
    w = out[dom].get("WRITE_SIZE", {}).get("per_launch_mean", 0.0)
    res["dominant_kernel"] = dom
    if have_fetch:
        res["dominant_kernel_hbm_bytes_per_launch"] = (f + w) * 1024.0
print(json.dumps(res, indent=1))
