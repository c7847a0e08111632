"""Condense a rocprofv3 --kernel-trace --stats run of bench.py into the text summary committed under profiles/:
whole-run stats of the hand-written kernels (from *_kernel_stats.csv) + steady-state breakdown (trace window)."""
import csv, sys, subprocess, os
stats, trace, warm, out = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
OURS = ("gagm_", "affinity_", "sinkhorn_", "mha_adj", "perm_loss", "node_", "sgd_multi", "gemm_f32", "gemm_splitk", "colsum", "nms_", "roi_align",
        "lap_batched", "rpn_decode", "box_inference", "paste_masks", "bias_act", "relu_bwd", "nchw_to_nhwc", "mask_pair_counts", "pair_stage", "gemm_grouped", "resize_", "row_scale", "rpn_select", "mask_measures", "mm_kernel", "mm_reduce")
rows = list(csv.DictReader(open(stats)))
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 2 --no-ab --no-cpu-baseline  (MI355X, gfx950; trained-regime checkpoint, free-running)\n")
    f.write("# hand-written kernels (libttdg_mgm.so), whole run incl. warm-up; durations in ns\n")
    f.write("%-64s %7s %14s %12s\n" % ("kernel", "calls", "total_ns", "avg_ns"))
    for r in rows:
        if any(k in r["Name"] for k in OURS):
            f.write("%-64s %7s %14s %12.0f\n" % (r["Name"].split("(")[0][:64], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"])))
    f.write("\n# steady-state window (after warm-up), all kernels\n")
    f.write(subprocess.check_output([sys.executable, os.path.join(os.path.dirname(__file__), "trace_summary.py"), trace, warm, "30"]).decode())
