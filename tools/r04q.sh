#!/bin/bash
cd /root/repo; O=gpurun_out/r04q; mkdir -p $O
for B in 128 256; do for S in 1 2 4 8; do
  echo "== B $B S $S"; PNGPD_BENCH_B=$B PNGPD_TRAIN_SPLITS=$S PNGPD_PASSES=bn2,fwd_train,gather,bwd_d,bwd_e,reduce timeout 120 python tools/bench_pass.py 2>/dev/null | grep -v "^B "
done; done | tee $O/pass_splits.txt
