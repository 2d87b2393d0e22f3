#!/bin/bash
# Re-creates this round's profile files under gpurun_out/r3 (summaries only; copied into profiles/r3 by hand).
# Every GPU step has its own time limit and the script stops at the first failure.  Token steps are issued as
# plain launches under the profiler (STT_MI355X_NO_GRAPH=1): it crashes on replayed graphs (profiles/README.md).
# bash scripts/refresh_profiles_r3.sh [part ...]   parts: bench profdefault decode600 pmc600 decode512 pmc512 beam encoder logmel
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
parts=${@:-bench profdefault decode600 pmc600 beam encoder logmel}
cd /tmp
last() { ls $1 | tail -n 1; }
for part in $parts; do
case $part in
bench)
  timeout -k 10 560 python3 $R/bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err ;;
profdefault)
  STT_MI355X_NO_GRAPH=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --pipelined 0 --no-cpu-baseline --beam-windows 0 --logmel-steps 0 --steps 1 --warmup 0 > $O/bench_default_profiled.json 2> $O/bench_default_profiled.err
  cp $(last "$O/raw/*/*kernel_stats.csv") $O/bench_default_kernel_stats.csv
  python3 $R/scripts/prof_shapes.py $(last "$O/raw/*/*kernel_trace.csv") 40 > $O/bench_default_kernel_shapes.txt
  rm -rf $O/raw ;;
decode512|decode600)
  W=${part#decode}
  STT_MI355X_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/scripts/prof_decode.py $W 32 > $O/decode$W.log 2>&1
  ( grep "per token" $O/decode$W.log; python3 $R/scripts/prof_shapes.py $(last "$O/raw/*/*kernel_trace.csv") 40 ) > $O/decode${W}_kernel_shapes.txt
  rm -rf $O/raw ;;
pmc512|pmc600)
  W=${part#pmc}
  for c in FETCH_SIZE WRITE_SIZE; do
    STT_MI355X_NO_GRAPH=1 timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $O/raw_$c -- python3 $R/scripts/prof_decode.py $W 8 > $O/pmc${W}_$c.log 2>&1
  done
  python3 $R/scripts/pmc_hbm.py "scripts/prof_decode.py $W 8 (large-v3, $W windows: one encoder pass, 3 x 8 greedy tokens as plain launches)" $(last "$O/raw_FETCH_SIZE/*/*counter_collection.csv") $(last "$O/raw_WRITE_SIZE/*/*counter_collection.csv") > $O/decode${W}_pmc_hbm.csv
  rm -rf $O/raw_FETCH_SIZE $O/raw_WRITE_SIZE ;;
beam)
  STT_MI355X_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/scripts/time_beam.py large-v3 100 32 > $O/beam100.log 2>&1
  ( grep -E "beam 5|cross-attention" $O/beam100.log; python3 $R/scripts/prof_shapes.py $(last "$O/raw/*/*kernel_trace.csv") 30 ) > $O/beam100_kernel_shapes.txt
  rm -rf $O/raw
  for c in FETCH_SIZE WRITE_SIZE; do
    STT_MI355X_NO_GRAPH=1 timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $O/raw_$c -- python3 $R/scripts/time_beam.py large-v3 100 8 > $O/pmcbeam_$c.log 2>&1
  done
  python3 $R/scripts/pmc_hbm.py "scripts/time_beam.py large-v3 100 8 (100 windows x 5 beams, timestamp rules, plain launches)" $(last "$O/raw_FETCH_SIZE/*/*counter_collection.csv") $(last "$O/raw_WRITE_SIZE/*/*counter_collection.csv") attn_decoder_cross > $O/beam100_pmc_hbm.csv
  rm -rf $O/raw_FETCH_SIZE $O/raw_WRITE_SIZE ;;
encoder)
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/scripts/prof_encoder.py 128 > $O/enc128.log 2>&1
  ( grep "encoder pass" $O/enc128.log; python3 $R/scripts/enc_layer_summary.py $(last "$O/raw/*/*kernel_trace.csv") ) > $O/encoder128_layer_kernels.txt
  rm -rf $O/raw ;;
logmel)
  timeout -k 10 300 python3 $R/bench.py --workload logmel > $O/bench_logmel_unprofiled.json 2> $O/bench_logmel.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --workload logmel --no-cpu-baseline > $O/bench_logmel_profiled.json 2>> $O/bench_logmel.err
  cp $(last "$O/raw/*/*kernel_stats.csv") $O/logmel_1h_kernel_stats.csv
  rm -rf $O/raw
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $O/raw_p$i -- python3 $R/scripts/time_frontend.py 1 > $O/pmc_logmel_p$i.log 2>&1
    python3 $R/scripts/pmc_summary.py $(last "$O/raw_p$i/*/*counter_collection.csv") logmel_pair > $O/logmel_sq_p$i.csv
    rm -rf $O/raw_p$i
  done
  cat $O/logmel_sq_p1.csv $O/logmel_sq_p2.csv > $O/logmel_1h_pmc_sq.csv; rm -f $O/logmel_sq_p1.csv $O/logmel_sq_p2.csv ;;
esac
echo "part $part done"
done
