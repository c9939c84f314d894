#!/bin/bash
# Kernel-trace profile of one bench configuration (no PMC passes): rocprofv3 kernel stats + the plain bench line.
# usage: bash tools/profile_kt.sh <tag> <config> [extra bench args]   -> gpurun_out/<tag>/<config>_{kernel_stats.csv,bench.json}
set -e
TAG=$1; CFG=$2; shift; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --config $CFG $*"
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$CFG" -o kt --output-format csv -- $BENCH --no-cpu-baseline --no-host-window > "$OUT/${CFG}_bench_kt.json" 2> "$OUT/${CFG}_kt.err"
cd - > /dev/null
timeout -k 10 900 $BENCH > "$OUT/${CFG}_bench.json" 2> "$OUT/${CFG}_bench.err"
python3 - "$OUT" "$CFG" <<'PY'
import csv, glob, re, sys
out, cfg = sys.argv[1], sys.argv[2]
src = glob.glob(f"{out}/kt_{cfg}/*kernel_stats.csv")[0]
with open(src, newline="") as f, open(f"{out}/{cfg}_kernel_stats.csv", "w", newline="") as g:
    w = csv.writer(g, quoting=csv.QUOTE_ALL)
    for r in csv.reader(f):  # kernel names without their argument lists
        r[0] = re.sub(r"\(.*$", "", r[0].replace("void ", "").replace("cls::(anonymous namespace)::", ""))[:140]
        w.writerow(r)
PY
find "$OUT/kt_$CFG" -name "*kernel_trace.csv" -delete
find "$OUT/kt_$CFG" -name "*agent_info.csv" -delete
head -c 200 "$OUT/${CFG}_bench.json"; echo
