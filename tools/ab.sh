timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])"
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/r1b && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r1b
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r1b -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r1b/bench_kt.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r1b -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r1b -o write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r1b
