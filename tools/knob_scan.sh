#!/bin/bash
# bench.py under a few runtime / engine switches; prints utt/s and ms per call.  usage: knob_scan.sh  (edit the list below)
run() { echo "== $* $EXTRA"; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-grouped $EXTRA 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['ms_per_call'])"; }
for i in 1 2; do
EXTRA="" run OAVSR_SKINNY_WIDE_MIN=8192
EXTRA="" run OAVSR_SKINNY_WIDE_MIN=1024
done
EXTRA="--streams 1" run OAVSR_SKINNY_WIDE_MIN=8192
EXTRA="--streams 1" run OAVSR_SKINNY_WIDE_MIN=1024
