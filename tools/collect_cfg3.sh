#!/bin/bash
# cfg-3 (BASELINE.json config 3: 8 graphs x 256 nodes, "rocprof HBM/MFMA capture") profile collection on the GPU box:
# kernel-trace stats of one MGM3_unsup forward+backward loop, then the PMC passes the north_star asks for - FETCH_SIZE and
# WRITE_SIZE in separate passes (MI355X_MICROARCH.md), and the issue-level SQ counters (VALU / LDS / MFMA activity per wave cycle)
# in a third.  Counters run with --kernel-trace only (gpurun refuses --pmc combined with the other trace domains).
#   usage: tools/collect_cfg3.sh r04
exec < /dev/null
set -u
R=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out/$R
mkdir -p $OUT
CMD="python tools/bench_core.py rand256x256x256x256x256x256x256x256 5"
OURS="gagm_|affinity_|sinkhorn_|perm_loss|mha_adj|gemm_f32|gemm_splitk|colsum"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3prof -o cfg3 -- $CMD > $OUT/cfg3_step.log 2> $OUT/cfg3_prof.log
echo "cfg3 stats rc=$?"
C3=$(find $OUT/cfg3prof -name "*kernel_stats.csv" | head -1)
if [ -n "$C3" ]; then python tools/make_cfg3_summary.py "$C3" $OUT/cfg3_step.log $OUT/cfg3_rocprof_summary.txt; cat $OUT/cfg3_rocprof_summary.txt | head -30; fi
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$OURS" --output-format csv -d $OUT/c3_r -o cfg3 -- $CMD > /dev/null 2> $OUT/c3_r.log
echo "FETCH rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$OURS" --output-format csv -d $OUT/c3_w -o cfg3 -- $CMD > /dev/null 2> $OUT/c3_w.log
echo "WRITE rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "$OURS" --output-format csv -d $OUT/c3_sq -o cfg3 -- $CMD > /dev/null 2> $OUT/c3_sq.log
echo "SQ rc=$?"
PR=$(find $OUT/c3_r -name "*counter_collection.csv" | head -1)
PW=$(find $OUT/c3_w -name "*counter_collection.csv" | head -1)
PS=$(find $OUT/c3_sq -name "*counter_collection.csv" | head -1)
if [ -n "$PR" ] && [ -n "$PW" ]; then python tools/pmc_summary.py "$PR" "$PW" $OUT/cfg3_pmc.json > /dev/null; else echo "no FETCH/WRITE csv"; tail -5 $OUT/c3_r.log; fi
if [ -n "$PS" ]; then python tools/pmc_sq_summary.py "$PS" $OUT/cfg3_sq_pmc.json > /dev/null; else echo "no SQ csv"; tail -5 $OUT/c3_sq.log; fi
timeout -k 10 200 python tools/bench_cfg3.py > $OUT/cfg3_kernels.json 2> $OUT/cfg3_kernels.err
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; rm -rf $OUT/c3_r $OUT/c3_w $OUT/c3_sq $OUT/cfg3prof
ls $OUT
