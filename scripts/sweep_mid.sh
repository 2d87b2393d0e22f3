#!/bin/bash
# Token-step GEMMs at 512 rows under different tilings of gemm_mid_kernel (rows per block, fill target):
# kernel trace of a 2-layer large-v3 decoder, per-kernel averages.  bash scripts/sweep_mid.sh
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sweep_mid
mkdir -p $O
cd /tmp
for cfg in "base" "STT_MI355X_MID_TM=1" "STT_MI355X_MID_TM=2" "STT_MI355X_MID_TM=3" "STT_MI355X_MID_TM=4" "STT_MI355X_MID_FILL=128" "STT_MI355X_MID_FILL=512"; do
  tag=$(echo $cfg | tr '=' '_')
  if [ "$cfg" != "base" ]; then export $cfg; fi
  timeout -k 10 200 rocprofv3 --kernel-trace -d $O/raw_$tag -- python3 $R/scripts/prof_decode.py 512 32 large-v3 2 > $O/$tag.log 2>&1
  ( echo "== $cfg"; grep "per token" $O/$tag.log | tail -n 1; python3 $R/scripts/prof_summary.py $(ls $O/raw_$tag/*/*.db | tail -n 1) 40 | grep -E "gemm_mid|gemm_skinny|layernorm_row|self_kernel|cross_kernel" ) > $O/$tag.txt
  rm -rf $O/raw_$tag
  if [ "$cfg" != "base" ]; then unset ${cfg%%=*}; fi
done
cat $O/*.txt
