#!/bin/bash
# Which entry of a measured plan file costs the step time?  bench.py on several plan files (W2L_TUNE_CACHE), interleaved, three
# rounds: copy the files to compare into a directory INSIDE the repo (gpurun_out/ does not travel to the GPU box) and run
#   bash tools/probe/plan_cmp.sh DIR name1 name2 ...        (DIR/name.txt are the plan files)
# Used in round 5 to find the one weight-gradient plan (768 -> 896, split 3 with atomics) behind a +0.09 ms selection.
dir=$1; shift
for rep in 1 2 3; do for c in "$@"; do
  W2L_TUNE_CACHE=$dir/$c.txt python3 bench.py --defer-wgrad 4 --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$c', json.loads(l)['ms_per_step'])"
done; done
