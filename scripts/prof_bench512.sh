#!/bin/bash
# Kernel trace of the default benchmark (512 windows per engine call) under rocprofv3, summaries only
# (the raw trace is ~100 MB).  The token step is issued as plain launches (STT_MI355X_NO_GRAPH=1): the
# profiler dies with SIGSEGV when a cached graph of the 512-window step is replayed by a second
# stt_generate call (profiles/README.md); the kernels and their durations are the same either way.
set -e
export TMPDIR=/tmp STT_MI355X_NO_GRAPH=1 STT_BENCH_VERBOSE=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof512
mkdir -p $O
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --pipelined 0 --no-cpu-baseline --beam-windows 0 --logmel-steps 0 --steps 1 --warmup 0 > $O/bench512_profiled.json 2> $O/bench512_profiled.err
cp $(ls $O/raw/*/*kernel_stats.csv | tail -n 1) $O/bench512_kernel_stats.csv
python3 $R/scripts/prof_shapes.py $(ls $O/raw/*/*kernel_trace.csv | tail -n 1) 40 > $O/bench512_kernel_shapes.txt
rm -rf $O/raw
echo done
