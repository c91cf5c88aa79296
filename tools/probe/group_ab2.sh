mkdir -p gpurun_out/r5f
run() { # name, args, env...
  name=$1; shift; a=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic $a 2> gpurun_out/r5f/$name.err | tail -1 > gpurun_out/r5f/$name.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5f/$name.json'))
print('$name', d['ms_per_step'], d.get('defer_wgrad',{}).get('measured_ms_per_step'))
PY
}
run j0 "--model jasper10x5 --batch 16" W2L_WGRAD_GROUPS=0
run ja "--model jasper10x5 --batch 16" W2L_WGRAD_GROUPS=auto W2L_WGRAD_GROUPS_VERBOSE=1
run ja4 "--model jasper10x5 --batch 16" W2L_WGRAD_GROUPS=auto W2L_WGRAD_GROUP_MAX=5
run j0b "--model jasper10x5 --batch 16" W2L_WGRAD_GROUPS=0
run w16_0 "--batch 16" W2L_WGRAD_GROUPS=0
run w16_a "--batch 16" W2L_WGRAD_GROUPS=auto
run w8_0 "--batch 8" W2L_WGRAD_GROUPS=0
run w8_a "--batch 8" W2L_WGRAD_GROUPS=auto
grep -h "wgrad group" gpurun_out/r5f/ja.err | sort | uniq | cut -c1-400 | head -30
