# scratch driver for one gpurun call of this round (not part of the product)
set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_scheduler_gpu.py -q -m gpu -k "8pp or r16 or 64_row" 2>&1 | tail -4
bash tools/round_profiles.sh r04 2>&1 | tail -40
