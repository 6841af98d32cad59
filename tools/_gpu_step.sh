# scratch driver for one gpurun call of this round (not part of the product)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "8pp or 8p or persistent_kernel or swiglu or attention" 2>&1 | tail -8 > $O/pp_tests.txt; cat $O/pp_tests.txt
timeout 300 python tools/gemm_pp_bench.py --shapes qkv,fc1,gateup --grids 0 --rounds 3 > $O/pp_bench.txt 2>&1; cat $O/pp_bench.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['ms_per_call'], d['config'].get('ids_verified'), d['config'].get('env'))
print(json.dumps(d['roofline'])); print(json.dumps(d['roofline_sides'])); print(json.dumps(d['per_call_decode']))
for k,v in list(d['stages'].items())[:12]: print(k, v)
PY
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_bf16_emulation_gpu.py tests/test_scheduler_gpu.py tests/test_harness_ingest_gpu.py -q -m gpu --durations=12 2>&1 | tail -30 > $O/tests2.txt; cat $O/tests2.txt
