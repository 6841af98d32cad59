# scratch driver for one gpurun call of this round (not part of the product)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_scheduler_gpu.py tests/test_harness_ingest_gpu.py tests/test_rccl_smoke_gpu.py tests/test_bf16_emulation_gpu.py -v -m gpu --durations=12 > $O/tests2_full.txt 2>&1
grep -n "PASSED\|FAILED\|ERROR\|Fatal\|Segmentation\|passed\|failed" $O/tests2_full.txt | tail -60
grep -n "Fatal Python error" -A25 $O/tests2_full.txt | head -60
grep -n "s call" $O/tests2_full.txt | head -14
