#!/bin/bash
# encoder pass of 1, 2, 3, 4 windows: split-K residual projections (default) against none
for b in 1 2 3 4; do
  for f in 1 0; do
    if [ $f = 1 ]; then export STT_MI355X_ENC_SPLIT=1; else unset STT_MI355X_ENC_SPLIT; fi
    echo "B=$b forced=$f: $(timeout -k 10 120 python scripts/prof_encoder.py $b 2>/dev/null | tail -n 1)"
  done
done
