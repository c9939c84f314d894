timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
B() { timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
cp classeq2_amd/csrc/libclsplace.so /tmp/base.so
B new
cp tools/variants/prev.so classeq2_amd/csrc/libclsplace.so && B prev
cp /tmp/base.so classeq2_amd/csrc/libclsplace.so && B new
timeout -k 10 600 python tools/len_probe.py 2>&1 | grep "L=  250\|L=  500"
