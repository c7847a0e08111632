"""Bookkeeping (not product): after a full `pytest -m gpu` run on a fresh box (gpurun merges gpurun_out/ back), copy the suite's tail to
profiles/<round>_gpu_suite_<name>.txt and append the box's statistical figures (trajectory fractions of bound, census rank sums, pooled
exchangeability z) to profiles/<round>_boxes.json.   usage: record_box.py r05 box2 gpurun_out/r05_suite_b.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, name, suite = sys.argv[1], sys.argv[2], sys.argv[3]
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
tail = open(os.path.join(ROOT, suite)).read().splitlines()[-24:]
with open(os.path.join(ROOT, "profiles", "%s_gpu_suite_%s.txt" % (rnd, name)), "w") as f:
    f.write("# python -m pytest tests -x -q -m gpu --durations=4   (fresh MI355X box, HEAD %s)\n" % head + "\n".join(tail) + "\n")
go = os.path.join(ROOT, "gpurun_out")
t = json.load(open(os.path.join(go, "trajectory.json")))
c = json.load(open(os.path.join(go, "trained_census.json")))
e = json.load(open(os.path.join(go, "census_exchangeability.json")))
rec = dict(head=head, passed=[l for l in tail if " passed" in l or " failed" in l][-1:],
           trajectory_worst_fraction_of_bound=t.get("worst_fraction_of_bound"), additive_worst_fraction=t.get("additive_worst_fraction"),
           dice_device=t.get("dice_device"), dice_host=t.get("dice_host"), outside_frozen_bound=t.get("outside_frozen_bound"),
           census=dict(rank_sum_z_objective=c["rank_sum_z_objective"], rank_sum_z_loss=c["rank_sum_z_loss"],
                       outside_reference_min_max=c["outside_reference_min_max"], min_max_checks=c["min_max_checks"],
                       outside_round4_bound=c["outside_round4_bound"], stage_states_defined=c["stage_states_defined_by_the_reference"],
                       stage_states_worst=c["stage_states_worst_device_deviation"]),
           exchangeability={k: dict(z_obj=v["pooled_z_objective"], z_loss=v["pooled_z_loss"]) for k, v in e["by_eps"].items()})
path = os.path.join(ROOT, "profiles", "%s_boxes.json" % rnd)
boxes = json.load(open(path)) if os.path.exists(path) else {}
boxes[name] = rec
zs = [b["census"]["rank_sum_z_objective"] for b in boxes.values() if isinstance(b, dict) and "census" in b]
zl = [b["census"]["rank_sum_z_loss"] for b in boxes.values() if isinstance(b, dict) and "census" in b]
boxes["pooled"] = dict(boxes=len(zs), mean_z_objective=sum(zs) / len(zs), mean_z_loss=sum(zl) / len(zl),
                       note="mean of the per-box rank-sum z over the fresh boxes of the round (every box fits its own checkpoint: independent draws); "
                            "an exchangeable solver has mean 0 with standard error >= 1 / sqrt(boxes)")
json.dump(boxes, open(path, "w"), indent=1)
print(name, rec["passed"], "census z", rec["census"]["rank_sum_z_objective"], rec["census"]["rank_sum_z_loss"], "| pooled", boxes["pooled"]["mean_z_objective"], boxes["pooled"]["mean_z_loss"])
