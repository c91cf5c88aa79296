#!/bin/bash
# Regenerates the artefacts committed under profiles/ (run through gpurun; outputs land in gpurun_out/final/ and are then
# copied into profiles/ with the round prefix by hand -- delete the local gpurun_out/final first (gpurun MERGES: files of older
# runs stay there): `rm -rf gpurun_out/final; gpurun ... bash tools/make_profiles.sh; for f in gpurun_out/final/*; do cp $f profiles/r03_$(basename $f); done`).
# Stages are independent: a failing one leaves its file empty, the others still run.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
rm -rf $OUT && mkdir -p $OUT
export W2L_TUNE_CACHE=$PWD/$OUT/tune_cache.txt   # the first run measures, the profiled runs reuse its choices
last() { tail -1; }
# ---- headline (BASELINE config 2): bench line, rocprofv3 kernel statistics (overlapped = default, and serialised), PMC traffic
python3 bench.py --steps 20 --warmup 5 2>/dev/null | last > $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial-wgrad > $OUT/prof_serial.log 2>&1
for d in default serial; do
  f=$(ls $OUT/prof_$d/*/*kernel_stats.csv | head -1)
  cp $f $OUT/${d}_kernel_stats.csv
  python3 tools/prof_summary.py stats $f > $OUT/${d}_kernel_families.txt
done
# the overlapped step as a timeline: MFMA-busy time, time with only HBM-bound kernels running, idle (tools/timeline.py) --
# once from the rocprofv3 trace (the host is ~3x slower under the profiler: bubbles at the step boundary that the real run
# does not have) and once from HIP events around every launch with no profiler attached (bench.py --event-trace)
python3 tools/timeline.py $(ls $OUT/prof_default/*/*kernel_trace.csv | head -1) --steps 6 --from 4 > $OUT/step_timeline.txt 2>&1
for spec in 0 6; do
  python3 bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-live-traffic --defer-wgrad=$spec --event-trace $OUT/event_trace_defer$spec.csv 2> /dev/null
  python3 tools/timeline.py $OUT/event_trace_defer$spec.csv --steps 4 --from 1 --gaps > $OUT/step_timeline_events_defer$spec.txt 2>&1
  rm -f $OUT/event_trace_defer$spec.csv
done
# same-box A/B of this round's two step-level changes: deferred weight gradients (0 vs 6 top units) and the three-tap AGPR
# weight-gradient kernels (W2L_WGRAD_NO_TAPS3=1 keeps them out of the measured selection; separate tune caches)
for rep in 1 2; do
  for v in "0 1" "0 0" "6 1" "6 0"; do
    set -- $v
    W2L_TUNE_CACHE=$PWD/$OUT/tune_ab_$2.txt $( [ "$2" = 1 ] && echo env W2L_WGRAD_NO_TAPS3=1 ) python3 bench.py --no-cpu-baseline --no-live-traffic --defer-wgrad=$1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('defer_wgrad=$1 no_taps3=$2 ms_per_step', d['ms_per_step'], 'wgrad TFLOP/s (serialised pass)', d['roofline']['wgrad_kernel']['achieved'])"
  done
done > $OUT/step_ab.txt 2>&1
rm -f $OUT/tune_ab_0.txt $OUT/tune_ab_1.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_bench_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 tools/prof_summary.py pmc $OUT/pmc_bench.json $OUT/pmc_bench_FETCH_SIZE $OUT/pmc_bench_WRITE_SIZE
rm -rf $OUT/pmc_bench_FETCH_SIZE $OUT/pmc_bench_WRITE_SIZE $OUT/prof_default $OUT/prof_serial $OUT/prof_default.log $OUT/prof_serial.log
# ---- one layer under the SQ / TCC counters (640 -> 640 k21)
bash tools/pmc_conv.sh 11 > $OUT/pmc_layer11.txt 2>&1
cp gpurun_out/pmc_11/summary.json $OUT/pmc_layer11.json
# ---- per-layer conv kernels, Wav2Letter table shapes at N = 32 (config 2) and N = 16 (the shapes of Jasper 10x5, config 4)
python3 tools/bench_conv.py --tune --fp8 > $OUT/conv_layers.txt 2>&1
python3 tools/bench_conv.py --tune --n 16 > $OUT/conv_layers_n16.txt 2>&1
# ---- the other workloads
python3 bench.py --model jasper10x5 --batch 16 --steps 10 --warmup 4 2>/dev/null | last > $OUT/bench_jasper10x5.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_jasper -- python3 bench.py --model jasper10x5 --batch 16 --steps 8 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python3 tools/prof_summary.py stats $(ls $OUT/prof_jasper/*/*kernel_stats.csv | head -1) > $OUT/jasper10x5_kernel_families.txt
rm -rf $OUT/prof_jasper
python3 bench.py --dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_fp8.json
python3 bench.py --model jasper10x5 --batch 16 --dtype fp8 --steps 10 --warmup 4 2>/dev/null | last > $OUT/bench_jasper10x5_fp8.json
python3 bench.py --model jasper10x5 --batch 16 --frames 16000 --dtype fp8 --steps 3 --warmup 2 2>/dev/null | last > $OUT/bench_jasper10x5_fp8_T16000.json
python3 bench.py --model jasper10x5 --batch 16 --frames 16000 --steps 3 --warmup 2 2>/dev/null | last > $OUT/bench_jasper10x5_T16000.json
python3 bench.py --mid-layers 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_w2l_mid1.json
for n in 8 16; do python3 bench.py --batch $n --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_w2l_n$n.json; done
# one rank, RCCL path forced: what the gradient collectives cost when nothing has to cross a link (plumbing, exposed_comm_ms)
python3 bench.py --force-dp --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_force_dp_1rank.json
# the multi-rank creation order rehearsed on one GPU (communicator before the first step), with and without the measured
# choice of side streams, and the same through the C ABI's RCCL helpers; the probe matrix of a fresh process
python3 bench.py --force-dp --early-collective --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_force_dp_comm_first.json
W2L_STREAM_PROBE=0 python3 bench.py --force-dp --early-collective --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_force_dp_comm_first_noprobe.json
W2L_DP_NATIVE=1 python3 bench.py --force-dp --early-collective --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_force_dp_native_rccl.json
python3 tools/stream_map.py 12 2>/dev/null > $OUT/stream_map.txt
python3 tools/bench_features.py 2>/dev/null | last > $OUT/bench_features.txt
cut -c1-1200 $OUT/bench.json
