#!/bin/bash
# A/B of kernel variants (tools/build_variant.sh) on the headline config: one line per variant.
# usage: bash tools/ab_bench.sh <out dir> <config> <variant[:ENV=VAL,...]> ...
OUT=$1; CFG=$2; shift; shift
mkdir -p "$OUT"
for spec in "$@"; do
  tag=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  lib=""
  if [ "$tag" != "default" ]; then lib="CLS_PLACE_LIB=$PWD/classeq2_amd/csrc/libclsplace_$tag.so"; fi
  name=$(echo "$spec" | tr ':=,' '___')
  env $lib $envs timeout -k 10 200 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-host-window > "$OUT/ab_$name.json" 2> "$OUT/ab_$name.err"
  python - "$OUT/ab_$name.json" "$spec" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:40s} value {d['value']/1e6:8.2f} M/s  step {d['ms_per_step']:7.3f} ms  kernel {r['kernel_ms']:7.3f} ms  {r['kernel']}", flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
done
