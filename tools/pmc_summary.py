"""Per-launch HBM traffic of the hand-written kernels from two rocprofv3 PMC passes (MI355X_MICROARCH.md: separate passes;
FETCH_SIZE / WRITE_SIZE count KB; on gfx950 FETCH_SIZE reports half of a wide coalesced read, so streaming kernels get x2).
usage: pmc_summary.py <FETCH counter_collection.csv> <WRITE counter_collection.csv> <out.json>"""
import csv, json, statistics, sys
OURS = ("gagm_kernel", "sgd_multi_tensor", "affinity_fwd", "affinity_bwd_kernel", "sinkhorn_pairs_fwd", "sinkhorn_pairs_bwd",
        "perm_loss_pair", "node_gather", "node_labels", "roi_align_fwd", "gagm_large_mul", "gagm_large_project", "mask_pair_counts",
        "roi_align_ml", "roi_align_nhwc", "roi_align_sep", "nchw_to_nhwc", "paste_masks", "bias_act", "relu_bwd", "mha_adjacency", "gemm_f32",
        "pair_stage_fwd", "pair_stage_bwd", "gemm_grouped", "resize_", "row_scale_multi", "rpn_select", "mask_measures", "mm_kernel", "mm_reduce")


def collect(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for k in OURS:
            if k in name:
                out.setdefault(k, []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


rd, wr = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(rd) | set(wr)):
    f = statistics.median(v for v, _ in rd.get(k, [(0, 0)]))
    w = statistics.median(v for v, _ in wr.get(k, [(0, 0)]))
    res[k] = {"launches": len(rd.get(k, [])), "FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w,
              "median_duration_us": statistics.median(d for _, d in rd.get(k, [(0, 0)])) / 1e3,
              "hbm_bytes_per_launch_raw": (f + w) * 1024.0,
              "hbm_bytes_per_launch_streaming_corrected": (2 * f + w) * 1024.0,
              # kernels launched at many sizes (bias_act, relu_bwd): the MEAN over the launches is what compares with the mean
              # algorithmic bytes per launch of the bench line
              "hbm_bytes_per_launch_mean_raw": (statistics.mean(v for v, _ in rd.get(k, [(0, 0)])) + statistics.mean(v for v, _ in wr.get(k, [(0, 0)]))) * 1024.0,
              "hbm_bytes_per_launch_mean_streaming_corrected": (2 * statistics.mean(v for v, _ in rd.get(k, [(0, 0)])) + statistics.mean(v for v, _ in wr.get(k, [(0, 0)]))) * 1024.0}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
