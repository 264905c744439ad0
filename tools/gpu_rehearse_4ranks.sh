#!/bin/bash
# Rehearsal, NOT a measurement: bench.py --gpus 4 with the four ranks sharing the one card of a gpurun box over gloo (host-side
# collectives), to run the N = 4 control flow (launcher, partition of the reducer's plan, captured graphs + collectives between them)
# on the real kernels. Times are meaningless (four processes on one GPU).
cd "$GRAFT_REPO_ROOT"
export EVP_BENCH_SHARE_DEVICE=1
for c in ${CONFIGS:-vit_base_rec vit_base_con swin_tiny_rec}; do
  timeout -k 10 280 python3 bench.py --gpus 4 --dist-backend gloo --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing \
      > gpurun_out/r4_rehearse4_$c.json 2> gpurun_out/r4_rehearse4_$c.err || { echo "$c FAILED"; tail -20 gpurun_out/r4_rehearse4_$c.err; exit 1; }
  tail -1 gpurun_out/r4_rehearse4_$c.json | cut -c1-400
done
