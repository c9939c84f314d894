timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout -k 10 300 python tools/poly_probe.py 35 2>&1 | tail -1
CLS_NO_FAST=1 timeout -k 10 300 python tools/poly_probe.py 35 2>&1 | tail -1
timeout -k 10 300 python tools/poly_probe.py 20 2>&1 | tail -1
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
