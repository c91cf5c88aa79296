mkdir -p gpurun_out/r5h
run() { # name, args, env...
  name=$1; shift; a=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic $a 2> gpurun_out/r5h/$name.err | tail -1 > gpurun_out/r5h/$name.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5h/$name.json'))
print('$name', d['ms_per_step'], d.get('defer_wgrad',{}).get('measured_ms_per_step'), d['config'].get('loss'))
PY
}
export W2L_WGRAD_GROUPS=0
run f0 "--defer-wgrad 6" W2L_FOLD_BN_FWD=0
run f1 "--defer-wgrad 6" W2L_FOLD_BN_FWD=1
run f0b "--defer-wgrad 6" W2L_FOLD_BN_FWD=0
run f1b "--defer-wgrad 6" W2L_FOLD_BN_FWD=1
run f1s4 "--defer-wgrad 6" W2L_FOLD_BN_FWD=1 W2L_STAT_SLOTS=4
run f1s16 "--defer-wgrad 6" W2L_FOLD_BN_FWD=1 W2L_STAT_SLOTS=16
run j0 "--model jasper10x5 --batch 16 --defer-wgrad 4" W2L_FOLD_BN_FWD=0
run j1 "--model jasper10x5 --batch 16 --defer-wgrad 4" W2L_FOLD_BN_FWD=1
