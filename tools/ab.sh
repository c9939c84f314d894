timeout -k 10 800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in "A=1"; do env $v timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['index'])"; done
