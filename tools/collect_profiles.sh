#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   bench line (default run), rocprofv3 kernel statistics of the same command, PMC HBM traffic (two separate passes),
#   per-shape contraction rates, the vgg 1 / vgg 5 / config 5 / waveform-in bench lines, the config-4 decode line.
# usage: bash tools/collect_profiles.sh r02_a
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench done"; cut -c1-300 $OUT/bench.json
# every profiler pass must EXIT 0: the CU-masked streams are destroyed by an atexit hook of src/hipabi.py (round 2's passes
# ended in a SIGSEGV inside __cxa_finalize with those streams alive); a non-zero exit stops the collection
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o x -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err \
  || { echo "rocprofv3 kernel-trace pass exited $?"; tail -5 $OUT/prof.err; exit 1; }
cp $(find $OUT/prof -name "x_kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv && echo "kernel stats done (profiler exit 0)" || { tail -5 $OUT/prof.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_bench.json 2> $OUT/pmc_fetch.err \
  || { echo "rocprofv3 FETCH_SIZE pass exited $?"; tail -5 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err \
  || { echo "rocprofv3 WRITE_SIZE pass exited $?"; tail -5 $OUT/pmc_write.err; exit 1; }
python3 tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "x_counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "x_counter_collection.csv" | head -1) $OUT/pmc_traffic.json $OUT/pmc_bench.json > /dev/null && echo "pmc done (both profiler passes exit 0)"
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write
python3 tools/bench_gemm16.py --json $OUT/gemm16_shapes.json > $OUT/gemm16_shapes.txt 2>&1 && echo "gemm shapes done"
python3 bench.py --vgg 1 --tokens 100 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_vgg1.json 2> $OUT/bench_vgg1.err && echo "vgg1 done"
python3 bench.py --vgg 5 --tokens 100 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_vgg5.json 2> $OUT/bench_vgg5.err && echo "vgg5 done"
python3 bench.py --waveform --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_waveform.json 2> $OUT/bench_waveform.err && echo "waveform done"
python3 bench.py --batch 64 --frames 3000 --tokens 400 --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_config5.json 2> $OUT/bench_config5.err && echo "config5 done"
python3 bench.py --shape librispeech --steps 16 --warmup 8 --no-cpu-baseline > $OUT/bench_librispeech_shape.json 2> $OUT/bench_librispeech_shape.err && echo "librispeech-shaped buckets done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof5 -o x -- python3 bench.py --batch 64 --frames 3000 --tokens 400 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/prof5.err \
  || { echo "rocprofv3 config-5 pass exited $?"; tail -5 $OUT/prof5.err; exit 1; }
cp $(find $OUT/prof5 -name "x_kernel_stats.csv" | head -1) $OUT/bench_config5_kernel_stats.csv && rm -rf $OUT/prof5 && echo "config5 kernel stats done"
python3 tools/bench_decode.py > $OUT/decode_config4.json 2> $OUT/decode_config4.err && echo "decode done"
python3 tools/bench_decode.py --utts 1 > $OUT/decode_config4_single.json 2>> $OUT/decode_config4.err
ls -la $OUT
