#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run; prints per-kernel averages of the counters named on the command
# line (tools/pmc_avg.py).  Usage: bash tools/pmc_pass.sh <tag> COUNTER [COUNTER ...]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=$1; shift
mkdir -p $OUT/prof
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" -d $OUT/prof/$TAG -o $TAG --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --render-frames 0 --fp32-mfma-steps 0 --batch8-steps 0 --steps 8 --warmup 6 > $OUT/prof/$TAG.log 2>&1
cd $ROOT
F=$(find $OUT/prof/$TAG -name '*counter_collection.csv' | head -1)
python3 tools/pmc_avg.py "$F" > $OUT/pmc_$TAG.txt
find $OUT/prof/$TAG -name '*counter_collection.csv' -delete
