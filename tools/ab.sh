cd classeq2_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -I../../include -DMIN_WAVES_PER_EU=1 cls_kernels.hip cls_api.cpp cls_db.cpp cls_fasta.cpp -o libclsplace.so 2>/dev/null
cd ../..
for B in 1 2 3 8; do
  CLS_BLOCKS_PER_CU=$B timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocks/CU=$B', d['roofline']['kernel_ms'])"
done
