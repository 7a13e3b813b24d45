#!/bin/bash
# BASELINE config 4 (8x8 = 64-filter bank, D = 192, batch 64 on one MI355X): kernel-trace stats of
# tools/stage_time.py plus separate --pmc passes for the MFMA-busy figure.
# Usage on the GPU box:  bash tools/profile_config4.sh r1
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
OUT=$R/gpurun_out/prof_c4_$TAG
rm -rf "$OUT"     # (gpurun merges gpurun_out/ across calls: never mix two runs' counter files)
mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/tools/stage_time.py 64 8 8 > $OUT/stage.log 2>&1
grep "^B=" $OUT/stage.log > $R/gpurun_out/profiles/${TAG}_config4_stage_time.txt || true
cp $OUT/trace/*/*kernel_stats.csv $R/gpurun_out/profiles/${TAG}_config4_kernel_stats.csv
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc/$tag -- python $R/tools/stage_time.py 64 8 8 > $OUT/pmc_$tag.log 2>&1
done
python $R/tools/pmc_summary.py $OUT/pmc > $R/gpurun_out/profiles/${TAG}_config4_pmc.txt
# HBM traffic: separate passes, never combined with other counters (FETCH_SIZE doubled on gfx950: tools/hbm_traffic.py)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/hbm/$c -- python $R/tools/stage_time.py 64 8 8 > $OUT/hbm_$c.log 2>&1
done
python $R/tools/hbm_traffic.py $OUT/hbm $R/gpurun_out/profiles/${TAG}_config4_hbm_traffic.json
cat $R/gpurun_out/profiles/${TAG}_config4_stage_time.txt $R/gpurun_out/profiles/${TAG}_config4_pmc.txt
