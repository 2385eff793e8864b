#!/bin/bash
# inverse kernel: with / without the exact tier, per output type and plane kind (interleaved in one process per line)
for kind in noise smooth; do
  for ot in f32 i16 u8; do
    echo "== kind=$kind out=$ot"
    python microbench/ab_forward.py inv=0x0 skipx=0x400 --kind $kind --direction inverse --out-type $ot --rounds 7
  done
done
python microbench/ab_forward.py fwd=0x1 fwd_skipx=0x401 --kind noise --rounds 7
