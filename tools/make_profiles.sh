#!/bin/bash
# Regenerates the artefacts committed under profiles/ (run through gpurun; outputs land in gpurun_out/final/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
mkdir -p $OUT
export W2L_TUNE_CACHE=$PWD/$OUT/tune_cache.txt   # first run measures, the profiled runs reuse its choices
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial-wgrad > $OUT/prof_serial.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_bench_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 tools/prof_summary.py pmc $OUT/pmc_bench.json $OUT/pmc_bench_FETCH_SIZE $OUT/pmc_bench_WRITE_SIZE
rm -rf $OUT/pmc_bench_FETCH_SIZE $OUT/pmc_bench_WRITE_SIZE
for d in default serial; do python3 tools/prof_summary.py stats $OUT/prof_$d/*/*kernel_stats.csv > $OUT/stats_$d.txt; done
bash tools/pmc_conv.sh 11 > $OUT/pmc_layer11.txt 2>&1
cp gpurun_out/pmc_11/summary.json $OUT/pmc_layer11.json
python3 bench.py --model jasper10x5 --batch 16 --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_jasper10x5.json
python3 bench.py --mid-layers 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_w2l_default_mid1.json
for n in 8 16; do python3 bench.py --batch $n --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_w2l_n$n.json; done
python3 tools/bench_features.py 2>/dev/null | tail -1 > $OUT/bench_features.txt
python3 tools/bench_conv.py --tune > $OUT/conv_layers.txt 2>&1
cat $OUT/bench.json | cut -c1-1200
