#!/bin/bash
# Kernel trace (start / end time stamps) of a short bench run, summarised into a per-queue timeline of one step.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/prof/trace -o trace --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --render-frames 0 --fp32-mfma-steps 0 --batch8-steps 0 --steps 10 --warmup 25 > $OUT/prof/trace.log 2>&1
cd $ROOT
F=$(find $OUT/prof/trace -name '*kernel_trace.csv' | head -1)
head -2 "$F" > $OUT/trace_head.txt
# the run ends with 10 steps that bracket every call with events (bench.py's kernel_ms leg): the step shown is one of
# the timed region before them
python3 tools/timeline.py "$F" $OUT/${1:-timeline}.txt -14 > /dev/null
find $OUT/prof/trace -name '*kernel_trace.csv' -delete
