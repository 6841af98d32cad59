#!/bin/bash
# A/B runs of bench.py under environment / command-line switches.  One configuration per line on stdin: "VAR=val VAR2=val | extra args"
# (either side may be empty); prints utt/s and ms per call for each.  Example:
#   printf 'OAVSR_SKINNY_PERSIST=0 |\nOAVSR_SKINNY_PERSIST=1 |\n | --streams 1\n' | bash tools/knob_scan.sh
while IFS='|' read -r envs extra; do
  [ -z "$envs$extra" ] && continue
  echo "== $envs | $extra"
  env $envs timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-grouped $extra 2>&1 | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['ms_per_call'])"
done
