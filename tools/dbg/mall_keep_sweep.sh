#!/bin/bash
# Round 6: how many bytes at the end of a sweep should be loaded WITHOUT the nontemporal hint (csrc/kmeans.hip, kp_nt_limit)?
# bench.py's steady state (4x6 bank, split-slab pass) and tools/stage_time.py (8x8 bank, deep-bank pass), two interleaved rounds.
# GCS_KP_MALL_MB: 0 = every load nt, 100000 = every load plain.   usage (GPU box): bash tools/dbg/mall_keep_sweep.sh
for round in 1 2; do
for mb in ${SWEEP:-100000 0 128 192 256}; do
  GCS_KP_MALL_MB=$mb timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4x6 keep_MB', $mb, d['value'], 'Mpix/s', d['ms_per_step'], 'ms')" || echo "$mb failed"
done
for mb in ${SWEEP:-100000 0 128 192 256}; do
  echo -n "8x8 keep_MB $mb "; GCS_KP_MALL_MB=$mb timeout -k 10 120 python tools/stage_time.py 64 8 8 2>/dev/null | tail -1 | sed 's/.*| pass/pass/; s/| in-step.*//'
done; done
