timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python tools/poly_probe.py 12 2>&1 | tail -2
