# scratch driver for one gpurun call of this round (not part of the product)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu --durations=15 ) > $O/full_suite.txt 2>&1; tail -25 $O/full_suite.txt
bash tools/pmc_r04.sh > $O/pmc.log 2>&1; tail -40 $O/pmc.log
