#!/bin/bash
# Regenerates every fixture in this directory. Needs /root/reference (this container only).
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
REF=/root/reference/BSD_metrics
export PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg
(cd $REF && /opt/conda/bin/python3.9 -W ignore $HERE/make_inputs.py $HERE/bsd_inputs.npz)
(cd $HERE/../.. && python $HERE/make_path_golden.py)
(cd $REF && /opt/conda/bin/python3.9 -W ignore $HERE/make_scoring_golden.py $HERE)
(cd $HERE && /opt/conda/bin/python3.9 -W ignore $HERE/make_bank_golden.py $HERE)
(cd $HERE && /opt/conda/bin/python3.9 -W ignore $HERE/make_feature_golden.py $HERE)
ls -la $HERE
