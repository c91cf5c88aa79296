#!/bin/bash
# PMC counters for the conv kernels on one layer shape (separate passes; no tracing domains besides kernel-trace)
# PMC_EXTRA=--fp8 adds the e4m3 weight-gradient kernel.  The library measures and picks its plans first (--tune), as the step engine
# does, so the counters are those of the kernels the step runs (the three-tap AGPR weight gradient on the k29 layers).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=${1:-11}
mkdir -p gpurun_out/pmc_$L
python3 tools/bench_conv.py --layers $L --reps 3 --tune --tune-cache gpurun_out/pmc_$L/plans.txt $PMC_EXTRA > /dev/null 2>&1   # measure the plans once, unprofiled
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$L/$tag -- python3 tools/bench_conv.py --layers $L --reps 3 --tune --tune-cache gpurun_out/pmc_$L/plans.txt $PMC_EXTRA > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_$L/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:48]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
summary = {}
for k, d in agg.items():
    if 'conv' not in k and 'w2l_wgrad3' not in k: continue
    print(k)
    name = ('conv_igemm_kernel' if 'igemm' in k else 'conv_wgrad_fp8_kernel' if 'wgrad_fp8' in k else
            k.split('(')[0] if 'w2l_wgrad3' in k else 'conv_wgrad_kernel')
    summary[name] = {c: sum(v) / len(v) for c, v in d.items()}
    for c, v in sorted(d.items()):
        print('   %-28s mean %.4g  (n=%d)' % (c, sum(v)/len(v), len(v)))
for name, d in summary.items():
    # FETCH_SIZE / WRITE_SIZE are KiB; gfx950 FETCH_SIZE counts wide streaming reads at 1/2 (MI355X_MICROARCH.md, HBM)
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        d['traffic_bytes_per_launch'] = 2 * d['FETCH_SIZE'] * 1024 + d['WRITE_SIZE'] * 1024
json.dump({'layer': $L, 'note': 'per-launch means over the launches of tools/bench_conv.py --layers $L', 'kernels': summary},
          open('gpurun_out/pmc_$L/summary.json', 'w'), indent=1)
PY
