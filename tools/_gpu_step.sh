# scratch driver for one gpurun call of this round (not part of the product)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_scheduler_gpu.py -q -m gpu -k "8pp or persistent_kernel or swiglu or 64_row" 2>&1 | tail -8 > $O/t1.txt; cat $O/t1.txt
timeout 300 python tools/gemm_pp_bench.py --shapes qkv,fc1,gateup --grids 0 --rounds 3 > $O/pp_bench.txt 2>&1; cat $O/pp_bench.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['ms_per_call'], d['config'].get('ids_verified'))
print(json.dumps(d['roofline'])); print(json.dumps(d['roofline_sides']));
for k,v in list(d['stages'].items())[:9]: print(k, v)
PY
