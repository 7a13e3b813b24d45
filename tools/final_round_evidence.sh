#!/bin/bash
# Everything a round's profiles/ set is built from, in ONE gpurun call (about 8 GPU-minutes):
#   gpurun --timeout 1200 -- 'bash tools/final_round_evidence.sh r5'
# GPU tests -> driver-style bench line -> round profile (kernel stats, HBM traffic) -> config-4 profile -> SQ counters ->
# scoring kernel stats -> one-image graph timeline -> steady-state Gabor stage timeline. Summaries land in gpurun_out/profiles/.
set -u
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
O=$R/gpurun_out; P=$O/profiles
mkdir -p $P
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/${TAG}_final_tests.log 2>&1
rc=$?; tail -2 $O/${TAG}_final_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed (rc=$rc): no further GPU step"; exit $rc; fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $P/${TAG}_bench_final.json 2> $O/${TAG}_final_bench.err || { echo "bench failed"; exit 1; }
timeout -k 10 300 bash tools/profile_round.sh $TAG > $O/${TAG}_final_prof.log 2>&1 || { echo "profile_round failed"; exit 1; }
timeout -k 10 300 bash tools/profile_config4.sh $TAG > $O/${TAG}_final_prof_c4.log 2>&1 || { echo "profile_config4 failed"; exit 1; }
timeout -k 10 300 bash tools/pmc_gabor.sh $TAG > $P/${TAG}_pmc.txt 2> $O/${TAG}_final_pmc.err || { echo "pmc failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_scoring $O/prof_small $O/prof_gstage
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_scoring -- python $R/tools/scoring_rate.py > $O/${TAG}_final_scoring.log 2>&1 \
  && cp $O/prof_scoring/*/*kernel_stats.csv $P/${TAG}_scoring_kernel_stats.csv
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof_small -- python $R/tools/small_call_probe.py > $O/${TAG}_final_small.log 2>&1 \
  && { echo "(rocprofv3 --kernel-trace of tools/small_call_probe.py, last replay of the one-image graph; under the tracer:"; grep "host->host" $O/${TAG}_final_small.log | tail -1; echo ")"; python $R/tools/small_step_timeline.py $O/prof_small; } > $P/${TAG}_small_call_timeline.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof_gstage -- python $R/tools/gabor_stage_probe.py steps > $O/${TAG}_final_gstage.log 2>&1 \
  && { python $R/tools/gabor_stage_probe.py show $O/prof_gstage; grep "gabor stage inside" $O/${TAG}_final_gstage.log | tail -1; } > $P/${TAG}_gabor_stage_timeline.txt 2>&1
ls $P | tr '\n' ' '
