B() { env $2 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
B key3 A=1
B key3 CLS_ORDER_BOTH_STRANDS=1
