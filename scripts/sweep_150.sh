#!/bin/bash
# Token-step GEMMs at 150 rows (what each of 8 ranks decodes per step of the 1 200-window job): K split and rows per
# block of gemm_mid_kernel, 2-layer large-v3 decoder under rocprofv3.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROWS=${1:-150}
O=$R/gpurun_out/sweep_$ROWS
mkdir -p $O
cd /tmp
for cfg in "base" "STT_MI355X_SPLIT_K1280=1" "STT_MI355X_SPLIT_K1280=2" "STT_MI355X_SPLIT_K1280=4" "STT_MI355X_SPLIT_K1280=5" "STT_MI355X_SPLIT_K5120=2" "STT_MI355X_SPLIT_K5120=4" "STT_MI355X_SPLIT_K5120=5" "STT_MI355X_SPLIT_K5120=8" "STT_MI355X_MID_TM=1" "STT_MI355X_MID_TM=2" "STT_MI355X_MID_TM=3"; do
  tag=$(echo $cfg | tr '=' '_')
  if [ "$cfg" != "base" ]; then export $cfg; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/raw_$tag -- python3 $R/scripts/prof_decode.py $ROWS 32 large-v3 2 > $O/$tag.log 2>&1
  ( echo "== $cfg"; grep "per token" $O/$tag.log | tail -n 1; python3 $R/scripts/prof_shapes.py $(ls $O/raw_$tag/*/*kernel_trace.csv | tail -n 1) 40 | grep -E "gemm_mid|gemm_skinny|layernorm_row" | grep -v "811x" ) > $O/$tag.txt
  rm -rf $O/raw_$tag
  if [ "$cfg" != "base" ]; then unset ${cfg%%=*}; fi
done
cat $O/*.txt | cut -c1-150
