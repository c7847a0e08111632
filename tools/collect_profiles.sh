#!/bin/bash
# Round-N profile collection on the GPU box (run through gpurun from the repo root): the bench line, the same command under
# rocprofv3 --kernel-trace --stats, and the two separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the same command.
# usage: tools/collect_profiles.sh r02
set -u
R=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/$R
mkdir -p $OUT
timeout 600 python bench.py --steps 16 --warmup 2 --no-ab --no-cpu-baseline > $OUT/bench_noab.json 2> $OUT/bench_noab.err   # also fills the checkpoint cache
echo "plain bench rc=$?"; ls -la /tmp/ttdg_synth_ckpt
CMD="python bench.py --steps 4 --warmup 2 --no-ab --no-cpu-baseline"
timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/prof.log; echo "stats pass rc=$?"
# counters only for the hand-written kernels: collecting them for every vendor kernel of a Mask R-CNN step serialises ~10^4
# dispatches per pass (the round-2 first attempt ran > 30 min)
OURS="gagm_|affinity_|sinkhorn_|sgd_multi|mask_pair|perm_loss|node_|roi_align|paste_masks|mha_adj|gemm_f32"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$OURS" -d $OUT/pmc_r -o bench -- $CMD > /dev/null 2> $OUT/pmc_r.log; echo "FETCH pass rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$OURS" -d $OUT/pmc_w -o bench -- $CMD > /dev/null 2> $OUT/pmc_w.log; echo "WRITE pass rc=$?"
tail -3 $OUT/prof.log
ST=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
TR=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/make_profile_summary.py $ST $TR 2 $OUT/bench_rocprof_summary.txt
head -41 $ST > $OUT/bench_kernel_stats_top40.csv
python tools/pmc_summary.py $(find $OUT/pmc_r -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_w -name "*counter_collection.csv" | head -1) $OUT/bench_pmc.json > /dev/null
rm -rf $OUT/prof/*kernel_trace.csv $OUT/pmc_r $OUT/pmc_w       # the raw traces exceed what travels back
ls -la $OUT
