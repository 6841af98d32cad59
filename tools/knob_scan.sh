#!/bin/bash
# bench.py under a few runtime / engine switches; prints utt/s and ms per call.  usage: knob_scan.sh  (edit the list below)
run() { echo "== $* $EXTRA"; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-grouped $EXTRA 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['ms_per_call'])"; }
EXTRA="" run A=1
EXTRA="" run OAVSR_NARROW_SPLIT=1
EXTRA="" run OAVSR_LORA_SPLIT_MAX=1
EXTRA="" run OAVSR_LORA_SPLIT_MAX=2
EXTRA="" run A=1
EXTRA="" run OAVSR_NARROW_SPLIT=1
EXTRA="--streams 1" run A=1
EXTRA="--streams 1" run OAVSR_NARROW_SPLIT=1
