#!/bin/bash
# K split of the residual / query projections of a token step at 600 rows (2-layer large-v3 decoder).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sweep_split
mkdir -p $O
cd /tmp
for cfg in "base" "STT_MI355X_SPLIT_K1280=1" "STT_MI355X_SPLIT_K1280=2" "STT_MI355X_SPLIT_K1280=4" "STT_MI355X_SPLIT_K5120=2" "STT_MI355X_SPLIT_K5120=4" "STT_MI355X_SPLIT_K5120=5"; do
  tag=$(echo $cfg | tr '=' '_')
  if [ "$cfg" != "base" ]; then export $cfg; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/raw_$tag -- python3 $R/scripts/prof_decode.py 600 32 large-v3 2 > $O/$tag.log 2>&1
  ( echo "== $cfg"; grep "per token" $O/$tag.log | tail -n 1; python3 $R/scripts/prof_shapes.py $(ls $O/raw_$tag/*/*kernel_trace.csv | tail -n 1) 40 | grep -E "gemm_mid|layernorm_row|cross_kernel<1" | grep -v "811x" ) > $O/$tag.txt
  rm -rf $O/raw_$tag
  if [ "$cfg" != "base" ]; then unset ${cfg%%=*}; fi
done
cat $O/*.txt | cut -c1-150
