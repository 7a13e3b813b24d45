#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the default bench.py run, (2) separate
# --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) for HBM traffic.
# Usage on the GPU box:  bash tools/profile_round.sh r1     -> gpurun_out/prof_r1/, summaries under profiles/
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"     # (gpurun merges gpurun_out/ across calls: never mix two runs' counter files)
mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-other-mode > $OUT/bench_trace.log 2>&1
grep '"metric"' $OUT/bench_trace.log > $R/gpurun_out/profiles/${TAG}_bench_under_rocprof.json || true
cp $OUT/trace/*/*kernel_stats.csv $R/gpurun_out/profiles/${TAG}_kernel_stats.csv
cp $R/gpurun_out/profiles/${TAG}_kernel_stats.csv $R/gpurun_out/profiles/kernel_stats_latest.csv   # bench.py: roofline.frac_rocprof
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-host-path --no-other-mode > $OUT/bench_$c.log 2>&1
done
# FETCH_SIZE factor of the split pass's access pattern (16 + 8 bytes per lane), from the streaming microbenchmark's known byte count
if [ -x $R/tools/ubench/read_stream ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- $R/tools/ubench/read_stream > $OUT/calib.log 2>&1 || true
  python $R/tools/hbm_calibrate.py $OUT/calib $R/gpurun_out/profiles/${TAG}_fetch_calibration.json || true
fi
python $R/tools/hbm_traffic.py $OUT $R/gpurun_out/profiles/${TAG}_hbm_traffic.json $R/gpurun_out/profiles/${TAG}_fetch_calibration.json
cp $R/gpurun_out/profiles/${TAG}_hbm_traffic.json $R/gpurun_out/profiles/hbm_traffic_latest.json
cat $R/gpurun_out/profiles/${TAG}_hbm_traffic.json
