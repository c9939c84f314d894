"""Copy one tools/profile.sh run (gpurun_out/<tag>/ + gpurun_out/<tag>_bench_full.json) into profiles/ as
<tag>_{pmc,kernel_stats,bench}_C3 and point profiles/traffic.json at it.  usage: finalize_profile.py <tag> [old_tag_to_remove]"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
d = json.load(open(os.path.join(src, "pmc_summary.json")))
d["workload"] = "C3: 10k-leaf tree, 1M x 150 bp reads, k=12, 1x MI355X"
d["note"] = ("FETCH_SIZE*1024 == TCC_MISS_sum*64 B here (narrow random reads: one 64-byte request per L2 miss), so the x2 "
             "streaming correction of MI355X_MICROARCH.md does not apply to this access pattern")
d["kernels"] = {(k if len(k) < 80 else k[:77] + "..."): v for k, v in d["kernels"].items()}
prof = os.path.join(root, "profiles")
json.dump(d, open(os.path.join(prof, f"{tag}_pmc_C3.json"), "w"), indent=1)
traffic = d["dominant_kernel_hbm_bytes_per_launch"]
json.dump({"C3:1000000": {"hbm_bytes_per_launch": traffic, "source": f"profiles/{tag}_pmc_C3.json", "kernel": d["dominant_kernel"]}},
          open(os.path.join(prof, "traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(prof, f"{tag}_kernel_stats_C3.csv"))
b = json.load(open(os.path.join(root, "gpurun_out", f"{tag}_bench_full.json")))
b["roofline"]["traffic"] = traffic  # (the line was printed with the previous traffic.json)
b["roofline"]["traffic_frac"] = traffic / (b["roofline"]["kernel_ms"] * 1e-3) / 1e9 / 8000.0
json.dump(b, open(os.path.join(prof, f"{tag}_bench_C3.json"), "w"))
if len(sys.argv) > 2:
    for suffix in ("pmc_C3.json", "kernel_stats_C3.csv", "bench_C3.json"):
        f = os.path.join(prof, f"{sys.argv[2]}_{suffix}")
        if os.path.exists(f):
            os.remove(f)
print(tag, b["value"], b["ms_per_step"], b["roofline"])
