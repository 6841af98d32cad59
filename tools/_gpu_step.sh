# scratch driver for one gpurun call of this round (not part of the product)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_scheduler_gpu.py tests/test_invariance_gpu.py tests/test_fullwidth_7b8b_gpu.py -q -m gpu -k "r16 or 64_row or invariance or regimes or fullwidth or 7b or 8b" 2>&1 | tail -12 > $O/t1.txt; cat $O/t1.txt
for cfg in "c4 2" "c4 3" "c4 4" "c3 3"; do set -- $cfg
  timeout 900 python bench.py --preset $1 --sched-lanes $2 --steps 20 --warmup 3 --no-cpu-baseline --no-grouped --no-roofline > $O/bench_$1_l$2.json 2> $O/bench_$1_l$2.err; echo "$cfg rc=$?"
  python - $O/bench_$1_l$2.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['value'], d['config']['ms_per_call'], d['config'].get('ids_verified'), json.dumps(d['config']['decode_scheduler']))
except Exception as e: print('ERR', e)
PY
done
