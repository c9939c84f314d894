timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout -k 10 300 python tools/poly_probe.py 2>&1 | grep -v amdgpu
CLS_FORCE_LIST=1 timeout -k 10 300 python tools/poly_probe.py 2>&1 | grep -v amdgpu
timeout -k 10 300 python tools/poly_probe.py 20 2>&1 | grep -v amdgpu
