#!/bin/bash
# One gpurun call: GPU tests, then (unless the tests were killed) the same-box A/B given as arguments.
# usage: tools/run_gpu_round.sh <tag> [ab.py arguments...]
set -u
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed (rc=$rc): no further GPU step"; exit $rc; fi
if [ $# -gt 0 ]; then
    timeout -k 10 300 python tools/ab.py "$@" > gpurun_out/${tag}_ab.log 2>&1
    echo "ab rc=$?"; tail -8 gpurun_out/${tag}_ab.log
fi
exit $rc
