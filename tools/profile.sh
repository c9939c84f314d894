#!/bin/bash
# Profiling recipe of this repo (run on the GPU box through gpurun): kernel trace + two PMC passes of bench.py.
# usage: bash tools/profile.sh <tag> [config] [extra bench args]  -> gpurun_out/<tag>/..., summaries for tools/finalize_profile.py
set -e
TAG=${1:-prof}
CFG=${2:-C3}
shift || true; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-window --config $CFG $*"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- $BENCH > "$OUT/bench_kt.json" 2> "$OUT/kt.err"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o fetch --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_tcc" -o tcc --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc_tcc.err"
cd - > /dev/null
timeout -k 10 300 $BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 tools/pmc_summary.py "$OUT" --command "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes) -- $BENCH" > "$OUT/pmc_summary.json"
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
src = glob.glob(sys.argv[1] + "/kt/*kernel_stats.csv")[0]
with open(src, newline="") as f, open(sys.argv[1] + "/kernel_stats.csv", "w", newline="") as g:
    w = csv.writer(g, quoting=csv.QUOTE_ALL)
    for r in csv.reader(f):  # kernel names without their argument lists
        r[0] = re.sub(r"\(.*$", "", r[0].replace("void ", "").replace("cls::(anonymous namespace)::", ""))[:140]
        w.writerow(r)
PY
rm -rf "$OUT"/pmc_fetch/*agent_info* "$OUT"/pmc_tcc/*agent_info*
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
python3 - "$OUT" "$CFG" <<'PY'
# stamp: which sources / kernel the PMC numbers belong to (bench.py only trusts a matching entry)
import json, sys
sys.path.insert(0, ".")
import bench
out, cfg = sys.argv[1], sys.argv[2]
d = json.load(open(out + "/pmc_summary.json"))
d["source_sha16"] = bench.source_sha16()
d["config"] = cfg
json.dump(d, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps({k: d.get(k) for k in ("dominant_kernel", "dominant_kernel_hbm_bytes_per_launch", "source_sha16")}))
PY
