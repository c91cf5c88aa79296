mkdir -p gpurun_out/r5e
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic"
run() { # name, env...
  name=$1; shift
  env "$@" $B 2> gpurun_out/r5e/$name.err | tail -1 > gpurun_out/r5e/$name.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5e/$name.json'))
print('$name', d['ms_per_step'], d.get('defer_wgrad'), d['roofline'].get('wgrad_kernel'))
PY
}
run base0 W2L_WGRAD_GROUPS=0 W2L_DEALT_WGRAD=0
run base W2L_WGRAD_GROUPS=0
run auto W2L_WGRAD_GROUPS=auto W2L_WGRAD_GROUPS_VERBOSE=1
run g3 "W2L_WGRAD_GROUPS=8,9,10;11,12,13" W2L_WGRAD_GROUPS_VERBOSE=1
run g3b "W2L_WGRAD_GROUPS=8,9,10;11,12,13;14,15,16,17,18,19" W2L_WGRAD_GROUPS_VERBOSE=1
run base0b W2L_WGRAD_GROUPS=0 W2L_DEALT_WGRAD=0
grep -h "wgrad group" gpurun_out/r5e/*.err | sort | uniq | head -20
