#!/bin/bash
# bench.py under two (or more) environment settings, interleaved, three rounds, one plan file: cross-process A/B of a switch that
# is read once per process.   bash tools/probe/env_ab.sh "VAR=a" "VAR=b" [-- bench args]
sets=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done; [ "$1" = "--" ] && shift
export W2L_TUNE_CACHE=/tmp/env_ab_plans.txt; rm -f $W2L_TUNE_CACHE
python3 bench.py --no-cpu-baseline --no-live-traffic "$@" > /dev/null 2>&1        # measures the plans once
for rep in 1 2 3; do for s in "${sets[@]}"; do
  env $s python3 bench.py --no-cpu-baseline --no-live-traffic "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$s', json.loads(l)['ms_per_step'])"
done; done
