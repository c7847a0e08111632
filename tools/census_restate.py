"""Offline restatement of the trained-regime solver census from the committed per-batch records
(profiles/r0*_trained_census*.json, gpurun_out/trained_census.json of the latest run): the rank-sum statistic of the device's
objective / loss against (a) all eight oracle members (round 3-5 gate), (b) the four ARITHMETIC-ONLY members - float32, float64,
inputs x (1 +- 1e-7 / 1e-6) - and (c) the four PROJECTION-NOISE members (noise multiplied into every Sinkhorn-stage projection:
annealed runs that reach better optima in the chaotic last stage); per batch: the round-4 bound, the 0.25-median floor, and the
floor min(arithmetic-only) - range(arithmetic-only).   usage: census_restate.py [out.json] [extra census json ...]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def midrank(value, samples):
    below = sum(1 for x in samples if x < value)
    ties = sum(1 for x in samples if x == value)
    return below + (ties + 2) / 2.0


def rank_sum_z(values, samples_per_step):
    n = len(values)
    if not n:
        return 0.0
    k = len(samples_per_step[0]) + 1
    tot = sum(midrank(v, s) for v, s in zip(values, samples_per_step))
    mean, var = n * (k + 1) / 2.0, n * (k * k - 1) / 12.0
    return (tot - mean) / var ** 0.5


def restate(path):
    d = json.load(open(path))
    weak = [r for r in d["records"] if not r["strong"]]
    out = {"file": os.path.relpath(path, ROOT), "batches": len(d["records"]), "weak": len(weak)}
    for name, sl in (("all8", slice(0, 8)), ("arith4", slice(0, 4)), ("noise4", slice(4, 8))):
        out["z_objective_" + name] = rank_sum_z([r["objective_device"] for r in weak], [r["objective_oracle_runs"][sl] for r in weak])
        out["z_loss_" + name] = rank_sum_z([r["loss_device"] for r in weak], [r["loss_oracle_runs"][sl] for r in weak])
    below_arith, below_r4, worst = 0, 0, 0.0
    for r in weak:
        a = r["objective_oracle_runs"][:4]
        floor = min(a) - (max(a) - min(a))
        below_arith += r["objective_device"] < floor
        below_r4 += bool(r.get("outside_round4_bound"))
        worst = max(worst, (min(a) - r["objective_device"]) / max(1e-12, max(a) - min(a)) if max(a) > min(a) else 0.0)
    out["below_arith_floor"], out["outside_round4_bound"], out["worst_deficit_in_arith_ranges"] = below_arith, below_r4, worst
    out["device_equals_oracle32"] = sum(bool(r.get("device_equals_oracle32")) for r in d["records"])
    return out


if __name__ == "__main__":
    outp = sys.argv[1] if len(sys.argv) > 1 else None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_trained_census*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0*_census_box*.json"))) + sys.argv[2:]
    rows = []
    for f in files:
        try:
            rows.append(restate(f))
        except (KeyError, ValueError) as e:
            print("skipped", f, repr(e))
    print("%-44s %4s | %17s | %17s | %17s | %s" % ("record", "weak", "z obj/loss all 8", "arithmetic-only 4", "projection-noise 4", "below arith floor / outside r4 bound / worst deficit (ranges)"))
    for r in rows:
        print("%-44s %4d | %+7.2f / %+7.2f | %+7.2f / %+7.2f | %+7.2f / %+7.2f | %d / %d / %.2f" % (
            r["file"], r["weak"], r["z_objective_all8"], r["z_loss_all8"], r["z_objective_arith4"], r["z_loss_arith4"], r["z_objective_noise4"], r["z_loss_noise4"],
            r["below_arith_floor"], r["outside_round4_bound"], r["worst_deficit_in_arith_ranges"]))
    n = len(rows)
    if n:
        mo, ml = sum(r["z_objective_arith4"] for r in rows) / n, sum(r["z_loss_arith4"] for r in rows) / n
        print("arithmetic-only pool: mean z objective %+.2f, loss %+.2f over %d records (2 / sqrt(n) = %.2f)" % (mo, ml, n, 2 / n ** 0.5))
    if outp:
        json.dump(rows, open(outp, "w"), indent=1)
