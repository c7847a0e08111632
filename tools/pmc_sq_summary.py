"""Per-kernel medians of every counter in a rocprofv3 counter_collection.csv (issue-level SQ counters of the hand-written
kernels).  usage: pmc_sq_summary.py <counter_collection.csv> <out.json>"""
import csv, json, statistics, sys
OURS = ("gagm_kernel", "gagm_large_mul", "gagm_large_project", "affinity_fwd", "affinity_bwd_kernel", "affinity_bwd_finish",
        "sinkhorn_pairs_fwd", "sinkhorn_pairs_bwd", "gemm_f32_kernel", "perm_loss_pair", "mha_adjacency", "sgd_multi_tensor",
        "bias_act", "mm_kernel", "roi_align_ml", "rpn_decode", "box_inference", "paste_masks", "nms_group")
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    for k in OURS:
        if k in name:
            d = acc.setdefault(k, {})
            d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            d.setdefault("_dur_ns", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            break
out = {}
for k, d in sorted(acc.items()):
    o = {c: statistics.median(v) for c, v in d.items() if c != "_dur_ns"}
    o["launches"] = len(next(iter(v for c, v in d.items() if c != "_dur_ns")))
    o["median_duration_us"] = statistics.median(d["_dur_ns"]) / 1e3
    wc, busy = o.get("SQ_WAVE_CYCLES"), o.get("SQ_BUSY_CYCLES")
    if wc:
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_VALU_MFMA_BUSY_CYCLES"):
            if c in o:
                o[c + "_per_wave_cycle"] = o[c] / wc
    out[k] = o
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
