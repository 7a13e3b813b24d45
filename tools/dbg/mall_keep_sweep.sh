#!/bin/bash
# Round 6: how many bytes at the end of a sweep should the Lloyd passes load WITHOUT the nontemporal hint (csrc/kmeans.hip,
# kp_nt_limit)? One variant build per budget (-DGCS_KP_MALL_KEEP_MB=n; 0 = every load nt, 100000 = every load plain), then
# bench.py's steady state (4x6 bank, split-slab pass) and tools/stage_time.py (8x8 bank, deep-bank pass), two interleaved rounds.
#   here:        bash tools/dbg/mall_keep_sweep.sh build            (build_ab/keep<n>.so)
#   GPU box:     bash tools/dbg/mall_keep_sweep.sh                  -> stdout (profiles/r6_mall_keep_sweep.txt was taken with an
#                                                                      environment hook of the same meaning, since removed)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SWEEP=${SWEEP:-100000 0 128 192 256 320}
if [ "${1:-}" = build ]; then
  for mb in $SWEEP; do bash $ROOT/tools/build_variant.sh keep$mb -DGCS_KP_MALL_KEEP_MB=$mb | tail -1; done
  exit 0
fi
cd $ROOT
for round in 1 2; do
for mb in $SWEEP; do
  timeout -k 10 120 python tools/bench_lib.py build_ab/keep$mb.so --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4x6 keep_MB', $mb, d['value'], 'Mpix/s', d['ms_per_step'], 'ms')" || echo "$mb failed"
done
for mb in $SWEEP; do
  echo -n "8x8 keep_MB $mb "; GCS_LIB_PATH=$ROOT/build_ab/keep$mb.so timeout -k 10 120 python tools/stage_time.py 64 8 8 2>/dev/null | tail -1 | sed 's/.*| pass/pass/; s/| in-step.*//'
done; done
