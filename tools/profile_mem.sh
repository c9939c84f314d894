#!/bin/bash
# L2 / fabric counters of one bench configuration (two PMC passes): lines fetched from the fabric, L2 hits and misses.
# usage: bash tools/profile_mem.sh <tag> <config> [extra bench args]  -> gpurun_out/<tag>/mem_<config>.json
set -e
TAG=$1; CFG=$2; shift; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT/mem_$CFG"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-host-window $*"
cd /tmp
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum -d "$OUT/mem_$CFG/a" -o a --output-format csv -- $BENCH > /dev/null 2> "$OUT/mem_$CFG/a.err"
timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$OUT/mem_$CFG/b" -o b --output-format csv -- $BENCH > /dev/null 2> "$OUT/mem_$CFG/b.err"
cd - > /dev/null
python3 tools/pmc_summary.py "$OUT/mem_$CFG" --command "rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum | --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum (separate passes) -- $BENCH" > "$OUT/mem_$CFG.json"
find "$OUT/mem_$CFG" -name "*counter_collection.csv" -delete
find "$OUT/mem_$CFG" -name "*agent_info.csv" -delete
python3 - "$OUT/mem_$CFG.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    if not k.startswith("place_fast_kernel<5"): continue
    print(k[:60], {c: round(x["per_launch_mean"] / 1e6, 2) for c, x in v.items()})
PY
