#!/bin/bash
# Round-N profile collection on the GPU box (run through gpurun from the repo root): the bench line, the same command under
# rocprofv3 --kernel-trace --stats, the two separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the same command restricted to
# the hand-written kernels, and the cfg-3 operator-level profile.   usage: tools/collect_profiles.sh r02
# Every step has its own time limit and reads nothing from stdin (a `head` on an empty file name once hung a whole call).
exec < /dev/null
set -u
R=${1:-r03}
export TMPDIR=/tmp
OUT=gpurun_out/$R
mkdir -p $OUT
timeout -k 10 600 python bench.py --steps 16 --warmup 2 --no-ab --no-cpu-baseline --no-cfg3 > $OUT/bench_noab.json 2> $OUT/bench_noab.err   # also fills the checkpoint cache
echo "plain bench rc=$?"
CMD="python bench.py --steps 4 --warmup 2 --no-ab --no-cpu-baseline --no-cfg3"
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/prof.log
echo "stats pass rc=$?"
OURS="gagm_|affinity_|sinkhorn_|pair_stage|sgd_multi|mask_pair|perm_loss|node_|roi_align|paste_masks|mha_adj|gemm_f32|gemm_grouped|bias_act|relu_bwd|resize_|row_scale|rpn_select|mask_measures|mm_kernel|mm_reduce"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$OURS" --output-format csv -d $OUT/pmc_r -o bench -- $CMD > /dev/null 2> $OUT/pmc_r.log
echo "FETCH pass rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$OURS" --output-format csv -d $OUT/pmc_w -o bench -- $CMD > /dev/null 2> $OUT/pmc_w.log
echo "WRITE pass rc=$?"
ST=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
TR=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
if [ -n "$ST" ] && [ -n "$TR" ]; then
  python tools/make_profile_summary.py "$ST" "$TR" 2 $OUT/bench_rocprof_summary.txt
  head -41 "$ST" > $OUT/bench_kernel_stats_top40.csv
else
  echo "no kernel stats csv"; ls -R $OUT/prof | head
fi
PR=$(find $OUT/pmc_r -name "*counter_collection.csv" | head -1)
PW=$(find $OUT/pmc_w -name "*counter_collection.csv" | head -1)
if [ -n "$PR" ] && [ -n "$PW" ]; then python tools/pmc_summary.py "$PR" "$PW" $OUT/bench_pmc.json > /dev/null; else echo "no pmc csv"; fi
# cfg-3 (operator level, 8 graphs x 256 nodes)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3prof -o cfg3 -- python tools/bench_core.py rand256x256x256x256x256x256x256x256 5 > $OUT/cfg3_step.log 2> $OUT/cfg3_prof.log
echo "cfg3 pass rc=$?"
C3=$(find $OUT/cfg3prof -name "*kernel_stats.csv" | head -1)
if [ -n "$C3" ]; then python tools/make_cfg3_summary.py "$C3" $OUT/cfg3_step.log $OUT/cfg3_rocprof_summary.txt; fi
timeout -k 10 200 python tools/bench_cfg3.py > $OUT/cfg3_kernels.json 2> $OUT/cfg3_kernels.err
echo "cfg3 kernels rc=$?"
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; rm -rf $OUT/pmc_r $OUT/pmc_w       # raw traces exceed what travels back
ls -la $OUT
