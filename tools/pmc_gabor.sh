#!/bin/bash
# Collect SQ counters for the Gabor kernel (separate --pmc passes, kernel-trace only).
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
OUT=$R/gpurun_out/pmc_$1
rm -rf "$OUT"     # (gpurun merges gpurun_out/ across calls: never mix two runs' counter files)
mkdir -p "$OUT"
shift
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-mode "$@" > $OUT.$tag.log 2>&1 || { tail -5 $OUT.$tag.log; }
done
python $R/tools/pmc_summary.py $OUT
