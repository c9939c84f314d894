#!/bin/bash
# Instruction-mix / wait-cycle counters (SQ block, two PMC passes) of one bench configuration.
# usage: bash tools/profile_sq.sh <tag> <config> [extra bench args]  -> gpurun_out/<tag>/sq_<config>.json
set -e
TAG=$1; CFG=$2; shift; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT/sq_$CFG"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-host-window $*"
cd /tmp
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES -d "$OUT/sq_$CFG/a" -o a --output-format csv -- $BENCH > /dev/null 2> "$OUT/sq_$CFG/a.err"
timeout -k 10 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d "$OUT/sq_$CFG/b" -o b --output-format csv -- $BENCH > /dev/null 2> "$OUT/sq_$CFG/b.err"
cd - > /dev/null
python3 tools/pmc_summary.py "$OUT/sq_$CFG" --command "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES | --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY (separate passes) -- $BENCH" > "$OUT/sq_$CFG.json"
find "$OUT/sq_$CFG" -name "*counter_collection.csv" -delete
find "$OUT/sq_$CFG" -name "*agent_info.csv" -delete
python3 - "$OUT/sq_$CFG.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    if not k.startswith("place_"): continue
    print(k[:60], {c: round(x["per_launch_mean"] / 1e6, 1) for c, x in v.items()})
PY
