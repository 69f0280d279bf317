#!/bin/bash
# Collects the judged artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/stats   rocprofv3 --kernel-trace --stats of `bench.py` (default steps and warm-up, as the plain bench line)
#   gpurun_out/prof/fetch, gpurun_out/prof/write, gpurun_out/prof/sq, gpurun_out/prof/mfma   one PMC pass each
#                           (FETCH_SIZE, WRITE_SIZE, eight SQ counters: instruction counts and the split of the
#                           waves' cycles, the matrix-core counters, the LDS traffic counters)
#   gpurun_out/bench.json   the plain bench line (with the CPU baseline)
# Every GPU step has its own time limit and the chain stops at the first failure.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT/prof
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu-baseline --render-frames 0 --fp32-mfma-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/stats -o stats --output-format csv -- python3 $B > $OUT/prof/stats.log 2>&1
echo "stats pass done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof/fetch -o fetch --output-format csv -- python3 $B > $OUT/prof/fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof/write -o write --output-format csv -- python3 $B > $OUT/prof/write.log 2>&1
echo "write pass done"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $OUT/prof/sq -o sq --output-format csv -- python3 $B > $OUT/prof/sq.log 2>&1
echo "sq pass done"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/prof/mfma -o mfma --output-format csv -- python3 $B > $OUT/prof/mfma.log 2>&1
echo "mfma pass done"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_LDS_ATOMIC_BANDWIDTH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/prof/lds -o lds --output-format csv -- python3 $B > $OUT/prof/lds.log 2>&1
echo "lds pass done"
cd $ROOT
F=$(find $OUT/prof/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/prof/write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py "$F" "$W" $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
cp "$(find $OUT/prof/stats -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
S=$(find $OUT/prof/sq -name '*counter_collection.csv' | head -1)
python3 tools/sq_summary.py "$S" $OUT/kernel_stats.csv $OUT/pmc_sq.json > $OUT/pmc_sq.txt
M=$(find $OUT/prof/mfma -name '*counter_collection.csv' | head -1)
python3 tools/pmc_avg.py "$M" > $OUT/pmc_mfma.txt
L=$(find $OUT/prof/lds -name '*counter_collection.csv' | head -1)
python3 tools/lds_summary.py "$L" $OUT/kernel_stats.csv $OUT/pmc_lds.json > $OUT/pmc_lds.txt
# keep the merged-back directory small: the raw per-dispatch CSVs are not needed afterwards
find $OUT/prof -name '*counter_collection.csv' -delete
find $OUT/prof -name '*kernel_trace.csv' -delete
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
