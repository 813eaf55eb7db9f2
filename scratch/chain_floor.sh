#!/bin/bash
# Chain-latency floor of the batch-1 decode graph: the same captured launch sequence with the weight stream removed.
#   GBX_CHAIN_FLOOR=2 : every GEMV batch (and lm_head row) re-reads rows 0 / 1 -> same instructions, launches, prologues, reductions,
#                       attention, hand-offs; no HBM weight stream
#   GBX_CHAIN_FLOOR=1 : every wave stops after one ring round
# Usage (GPU box): bash scratch/chain_floor.sh > profiles/r03_chain_floor.txt
for fuse in 0 1; do
  for fl in 0 2 1; do
    for rep in 1 2 3; do
      if [ $fl = 0 ]; then unset GBX_CHAIN_FLOOR; else export GBX_CHAIN_FLOOR=$fl; fi
      GBX_FUSE_AO=$fuse python bench.py --no-extra --no-cpu-baseline --no-kernel-leg --steps 256 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('launches_per_layer=%d chain_floor_mode=%s run=%s  %.1f tok/s  %.1f us/token' % (5 - $fuse, '$fl', '$rep', d['value'], d['ms_per_step']*1000))"
    done
  done
done
