"""Copy one tools/profile.sh run (gpurun_out/<tag>/) into profiles/ as <tag>_{pmc,kernel_stats,bench}_<config> and
point profiles/traffic.json at it, stamped with the source fingerprint and the kernel instance the counters were
taken from (bench.py reports `traffic: null, traffic_stale: true` for anything else).
usage: finalize_profile.py <tag> [old_tag_to_remove]"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
src = os.path.join(root, "gpurun_out", tag)
d = json.load(open(os.path.join(src, "pmc_summary.json")))
cfg = d.get("config", "C3")
b = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1])
d["workload"] = b["config"]["workload"]
d["note"] = ("FETCH_SIZE*1024 == TCC_MISS_sum*64 B here (narrow random reads: one 64-byte request per L2 miss), so the x2 "
             "streaming correction of MI355X_MICROARCH.md does not apply to this access pattern")
d["kernels"] = {(k if len(k) < 80 else k[:77] + "..."): v for k, v in d["kernels"].items()}
prof = os.path.join(root, "profiles")
json.dump(d, open(os.path.join(prof, f"{tag}_pmc_{cfg}.json"), "w"), indent=1)
traffic = d["dominant_kernel_hbm_bytes_per_launch"]
tpath = os.path.join(prof, "traffic.json")
try:
    table = json.load(open(tpath))
except (OSError, ValueError):
    table = {}
reads = b["config"]["reads_per_gpu"]
table[f"{cfg}:{reads}"] = {"hbm_bytes_per_launch": traffic, "source": f"profiles/{tag}_pmc_{cfg}.json", "kernel": d["dominant_kernel"],
                           "source_sha16": d["source_sha16"]}
json.dump(table, open(tpath, "w"), indent=1)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(prof, f"{tag}_kernel_stats_{cfg}.csv"))
r = b["roofline"]
r["traffic"] = traffic  # (the line was printed before this profile was folded into traffic.json)
r["traffic_stale"] = False
ksec = r["kernel_ms"] * 1e-3
r["traffic_frac"] = traffic / ksec / 1e9 / 8000.0
if r.get("needed_bytes_per_launch"):
    r["overfetch"] = traffic / r["needed_bytes_per_launch"]
if isinstance(r.get("random_line_rate"), dict):
    g = traffic / 64.0 / ksec / 1e9
    r["random_line_rate"]["achieved_glines_per_s"] = g
    r["random_line_rate"]["frac"] = g / r["random_line_rate"]["peak_glines_per_s"]
json.dump(b, open(os.path.join(prof, f"{tag}_bench_{cfg}.json"), "w"))
if len(sys.argv) > 2:
    for suffix in (f"pmc_{cfg}.json", f"kernel_stats_{cfg}.csv", f"bench_{cfg}.json"):
        f = os.path.join(prof, f"{sys.argv[2]}_{suffix}")
        if os.path.exists(f):
            os.remove(f)
print(tag, cfg, b["value"], b["ms_per_step"], json.dumps(r))
