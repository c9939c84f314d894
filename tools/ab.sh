B() { env $1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['index'])"; }
B A=1 "--config C3"
B A=1 "--config C2"
timeout -k 10 300 python tools/poly_probe.py 35 2>&1 | tail -2
timeout -k 10 300 python tools/poly_probe.py 20 2>&1 | tail -1
