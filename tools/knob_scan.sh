#!/bin/bash
# bench.py under a few runtime / engine switches; prints utt/s and ms per call.  usage: knob_scan.sh  (edit the list below)
run() { echo "== $* $EXTRA"; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-grouped $EXTRA 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['ms_per_call'])"; }
for i in 1 2; do
EXTRA="" run OAVSR_EXP_PERSIST_SWIGLU=0
EXTRA="" run OAVSR_EXP_PERSIST_SWIGLU=1 OAVSR_EXP_PERSIST_PER_CU=1
EXTRA="" run OAVSR_EXP_PERSIST_SWIGLU=1
done
EXTRA="--streams 1" run OAVSR_EXP_PERSIST_SWIGLU=0
EXTRA="--streams 1" run OAVSR_EXP_PERSIST_SWIGLU=1 OAVSR_EXP_PERSIST_PER_CU=1
