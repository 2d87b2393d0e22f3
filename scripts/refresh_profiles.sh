#!/bin/bash
# Re-creates the round's profile files under gpurun_out/ (copied into profiles/ by hand afterwards).
# Every GPU step has its own time limit and the script stops at the first failure.
set -e
export TMPDIR=/tmp
O=gpurun_out/refresh
mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python bench.py --pipelined 0 --no-cpu-baseline --batch 256 > $O/bench_profiled.json 2> $O/bench_profiled.err
cp $(ls $O/bench_prof/*/*kernel_stats.csv | tail -n 1) $O/bench_default_kernel_stats.csv
rm -rf $O/bench_prof
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/dec_prof -- python scripts/prof_decode.py 256 32 > $O/decode256.log 2>&1
( grep "per token" $O/decode256.log; python scripts/prof_summary.py $(ls $O/dec_prof/*/*.db | tail -n 1) 32 ) > $O/decode256_kernel_shapes.txt
rm -rf $O/dec_prof
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/enc_prof -- python scripts/prof_encoder.py 128 > $O/enc128.log 2>&1
python scripts/enc_layer_summary.py $(ls $O/enc_prof/*/*kernel_trace.csv | tail -n 1) > $O/encoder128_layer_kernels.txt
rm -rf $O/enc_prof
echo done
