#!/bin/bash
# Regenerates the artefacts committed under profiles/ (run through gpurun; outputs land in gpurun_out/final/ and are then
# copied into profiles/ with the round prefix by hand -- delete the local gpurun_out/final first (gpurun MERGES: files of older
# runs stay there): `rm -rf gpurun_out/final; gpurun ... bash tools/make_profiles.sh; for f in gpurun_out/final/*; do cp $f profiles/r03_$(basename $f); done`).
# Stages are independent: a failing one leaves its file empty, the others still run.
# Four stages (a gpurun call is limited to 20 minutes): `bash tools/make_profiles.sh headline`, `... kernels`, `... workloads`,
# `... replay` (no argument: all four); each writes its own files under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
mkdir -p $OUT
STAGE=${1:-all}
export W2L_TUNE_CACHE=$PWD/$OUT/tune_cache_$STAGE.txt   # the first run measures, the profiled runs reuse its choices
last() { tail -1; }
if [ "$STAGE" = headline ] || [ "$STAGE" = all ]; then
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
# same-box, same-process, interleaved A/B of this round's step-level switches (tools/step_ab.py): grouped weight gradients, the
# two-launch BatchNorm-backward chain, where deferred launches go, the folded forward finalize, weight gradient after the data
# gradient, round 4's configuration as a whole -- and the what-if probes (no weight gradients at all / no optimizer step)
python3 tools/step_ab.py --rounds 6 --variants "base;nogroups=GROUPS:0;groups8=WGROUP_MAX:8;slowbn=FAST_BN_BWD:0;start=DEFER_SPREAD:start;foldfwd=FOLD_BN_FWD:1;late=WGRAD_AFTER_DGRAD:1;round4=GROUPS:0,FAST_BN_BWD:0,DEFER_SPREAD:start;base2" 2>/dev/null > $OUT/step_ab.txt
python3 tools/step_ab.py --rounds 4 --variants "base;nowgrad=PROBE:nowgrad,DEFER:0;nosgd=PROBE:nosgd,DEFER:0;defer0=DEFER:0" 2>/dev/null >> $OUT/step_ab.txt
python3 tools/step_ab.py --rounds 4 --batch 16 --variants "base;nogroups=GROUPS:0;groups8=WGROUP_MAX:8;slowbn=FAST_BN_BWD:0" 2>/dev/null >> $OUT/step_ab.txt
python3 tools/step_ab.py --rounds 4 --model jasper10x5 --batch 16 --defer 4 --variants "base;nogroups=GROUPS:0;groups8=WGROUP_MAX:8;slowbn=FAST_BN_BWD:0;round4=GROUPS:0,FAST_BN_BWD:0,DEFER_SPREAD:start" 2>/dev/null >> $OUT/step_ab.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_bench_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 tools/prof_summary.py pmc $OUT/pmc_bench.json $OUT/pmc_bench_FETCH_SIZE $OUT/pmc_bench_WRITE_SIZE
rm -rf $OUT/pmc_bench_FETCH_SIZE $OUT/pmc_bench_WRITE_SIZE $OUT/prof_default $OUT/prof_serial $OUT/prof_default.log $OUT/prof_serial.log
fi
if [ "$STAGE" = kernels ] || [ "$STAGE" = all ]; then
# ---- one layer under the SQ / TCC counters (640 -> 640 k21)
bash tools/pmc_conv.sh 11 > $OUT/pmc_layer11.txt 2>&1
cp gpurun_out/pmc_11/summary.json $OUT/pmc_layer11.json
# ---- per-layer conv kernels, Wav2Letter table shapes at N = 32 (config 2) and N = 16 (the shapes of Jasper 10x5, config 4)
python3 tools/bench_conv.py --tune --fp8 > $OUT/conv_layers.txt 2>&1
python3 tools/bench_conv.py --tune --n 16 > $OUT/conv_layers_n16.txt 2>&1
# every weight-gradient plan class per layer (best split of each; + 32: its dealt stream-K form) -- the numbers DESIGN 3 quotes
python3 tools/bench_conv.py --layers 5,6,7,8,9,10,11 --wgrad-plans --reps 20 --warm-s 1 > $OUT/wgrad_plans.txt 2>&1
# grouped weight-gradient launches against the same layers one by one (tools/bench_wgrad_group.py), and the table's weight
# gradients as the step would launch them with groups of at most three
python3 tools/bench_wgrad_group.py --groups "12,11,10;9,8,7;15,14,13;13,12,11;6,5,4;3,2,1;18,17;18,17,16;9,8,7,6,5,4,3,2" > $OUT/wgrad_groups.txt 2>&1
# the BatchNorm kernels alone: three-launch and two-launch backward chains, finalize + apply vs the folded forward kernel
python3 tools/bench_elem.py > $OUT/bn_kernels.txt 2>&1
fi
if [ "$STAGE" = workloads ] || [ "$STAGE" = all ]; then
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
python3 bench.py --ragged --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_ragged.json
fi
if [ "$STAGE" = replay ] || [ "$STAGE" = all ]; then
# ---- round 6: the reference's loop body and the recorded launch lists
# host lead / host ms per step (bench.py --lead-trace: last column pair), replayed and eager, headline and Jasper 10x5
python3 bench.py --lead-trace --steps 12 --warmup 8 --no-cpu-baseline 2> $OUT/lead_trace.txt > /dev/null
W2L_REPLAY=0 python3 bench.py --lead-trace --steps 12 --warmup 8 --no-cpu-baseline 2> $OUT/lead_trace_eager.txt > /dev/null
python3 bench.py --model jasper10x5 --batch 16 --lead-trace --steps 12 --warmup 8 --no-cpu-baseline 2> $OUT/lead_trace_jasper10x5.txt > /dev/null
W2L_REPLAY=0 python3 bench.py --model jasper10x5 --batch 16 --lead-trace --steps 12 --warmup 8 --no-cpu-baseline 2> $OUT/lead_trace_jasper10x5_eager.txt > /dev/null
# replayed vs eager steps interleaved in one process: device ms per step and host enqueue ms per step
for cfg in "" "--batch 8" "--batch 16" "--mid-layers 1" "--model jasper10x5 --batch 16" "--model jasper10x5 --batch 16 --dtype fp8" "--dtype fp8"; do
  echo "== tools/replay_ab.py $cfg"; python3 tools/replay_ab.py $cfg 2>/dev/null | head -3
done > $OUT/replay_ab.txt
# the timed region through training_step (host batch, per-step greedy decode + CER / WER): value / ms_per_step ARE the trainer's
python3 bench.py --through-trainer --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic 2>/dev/null | last > $OUT/bench_through_trainer.json
python3 bench.py --through-trainer --model jasper10x5 --batch 16 --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_through_trainer_jasper10x5.json
python3 bench.py --through-trainer --model jasper10x5 --batch 16 --dtype fp8 --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | last > $OUT/bench_through_trainer_jasper10x5_fp8.json
# SURVEY 8d's other CPU-baseline legs: the shipped default mid_layers: 1 at N = 32, Jasper at N = 2 (cpu_baseline of each line)
python3 bench.py --mid-layers 1 --steps 20 --warmup 5 2>/dev/null | last > $OUT/bench_w2l_mid1_cpu_baseline.json
python3 bench.py --model jasper10x5 --batch 16 --steps 6 --warmup 4 --no-trainer-leg 2>/dev/null | last > $OUT/bench_jasper10x5_cpu_baseline.json
fi
ls $OUT | head -80
