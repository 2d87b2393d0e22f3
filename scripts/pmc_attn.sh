#!/bin/bash
# SQ counter passes over the encoder attention kernel (scripts/prof_encoder.py 32: one encoder pass of 32 windows x 4).
export TMPDIR=/tmp
O=gpurun_out/pmc_attn
mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32" \
           "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_VMEM"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python scripts/prof_encoder.py 32 > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; continue; }
  python scripts/pmc_summary.py $(ls $O/p$i/*/*counter_collection.csv | tail -n 1) attn_encoder > $O/p$i.csv
  rm -rf $O/p$i
  cat $O/p$i.csv
done
