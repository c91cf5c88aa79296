mkdir -p gpurun_out/r5g
run() { # name, args, env...
  name=$1; shift; a=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic $a 2> gpurun_out/r5g/$name.err | tail -1 > gpurun_out/r5g/$name.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5g/$name.json'))
print('$name', d['ms_per_step'], d.get('defer_wgrad',{}).get('measured_ms_per_step'))
PY
}
export W2L_WGRAD_GROUPS=0
run start6 "--defer-wgrad 6" W2L_DEFER_SPREAD=start
run even6 "--defer-wgrad 6" W2L_DEFER_SPREAD=even
run la6_6 "--defer-wgrad 6" W2L_DEFER_SPREAD=la:6
run la4_6 "--defer-wgrad 6" W2L_DEFER_SPREAD=la:4
run even8 "--defer-wgrad 8" W2L_DEFER_SPREAD=even
run la5_8 "--defer-wgrad 8" W2L_DEFER_SPREAD=la:5
run even10 "--defer-wgrad 10" W2L_DEFER_SPREAD=even
run la5_10 "--defer-wgrad 10" W2L_DEFER_SPREAD=la:5
run start6b "--defer-wgrad 6" W2L_DEFER_SPREAD=start
run even4 "--defer-wgrad 4" W2L_DEFER_SPREAD=even
