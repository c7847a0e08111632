"""Is the device's solver exchangeable with the reference algorithm's own runs where the final answer is NOT well defined?
(VERDICT r4 "What's weak" 3 / "Next round" 1b.)

In the bench regime the last Sinkhorn stage of GA_GM.gagm (multi_graph_matching.py:300-389) is chaotic under rounding for the
reference itself: the in-suite census ranks ONE device answer among eight reference answers per batch, over the 16
DEPENDENT batches of one continual run, and its rank-sum statistic was not centred over six boxes (z_loss -0.5 ... +3.0).  This
study removes both weaknesses: on each of the 8 recorded trained-regime solver inputs (tools/fixtures/trained_solver_inputs.pt -
independent of the box, of the checkpoint fit and of each other) it draws
    N oracle solves   (oracle/gmodule.gagm, float32) on inputs perturbed by 1e-7-relative noise, seeds 5000 + 100 j + k
    N device solves   (ttdg_gagm_solve)               on inputs perturbed the same way,           seeds 7000 + 100 j + k
- independent draws on both sides - and compares the two samples of the objective <W, U U^T> and of the matching loss
(perm_loss_of, both evaluated on the UNPERTURBED W) with a two-sample Mann-Whitney test per input and a pooled statistic
(sum of the per-input U statistics, standardised: van Elteren with equal weights).  Under exchangeability the pooled z is N(0, 1).
It also records how many distinct answers (U U^T) each side produced per input.

usage (GPU box):  python tools/census_exchangeability.py [N=32] [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def perturbed(W, U0, seed, eps=1e-7):
    from ttdg_mgm_amd import synth
    g = synth.gen(seed)
    return W * (1 + eps * synth.normal(g, tuple(W.shape))), U0 * (1 + eps * synth.normal(g, tuple(U0.shape)))


def mann_whitney(x, y):
    """-> (U statistic of x, its null mean, its null variance with the tie correction)."""
    import scipy.stats
    n, m = len(x), len(y)
    r = scipy.stats.rankdata(np.concatenate([x, y]))
    U = float(r[:n].sum() - n * (n + 1) / 2.0)
    _, counts = np.unique(np.concatenate([x, y]), return_counts=True)
    tie = float(((counts ** 3 - counts).sum()) / ((n + m) * (n + m - 1)))
    var = n * m / 12.0 * ((n + m + 1) - tie)
    return U, n * m / 2.0, var


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "census_exchangeability.json")
    import admission
    from oracle import gmodule as og
    from ttdg_mgm_amd import ops
    dev = torch.device("cuda:0")
    inputs = torch.load(os.path.join(ROOT, "tools", "fixtures", "trained_solver_inputs.pt"), weights_only=True)
    rows, pooled = [], {"objective": [0.0, 0.0], "loss": [0.0, 0.0]}
    t0 = time.time()
    for j, rec in enumerate(inputs):
        sizes = [int(n) for n in rec["sizes"]]
        apack, W, U0 = rec["apack"].float(), rec["Wds"].float(), rec["U0"].float()
        A = admission.unpack_adjacency(apack, sizes)
        gr = ops.graphs(sizes)
        side = {"oracle": [], "device": []}
        for k in range(N):
            Wp, Up = perturbed(W, U0, 5000 + 100 * j + k)
            side["oracle"].append(og.gagm(A, Wp, Up, sizes))
            Wp, Up = perturbed(W, U0, 7000 + 100 * j + k)
            side["device"].append(ops.gagm_solve(apack.to(dev), Wp.to(dev), Up.to(dev), gr, sizes)[0].cpu())
        row = dict(input=j, sizes=sizes)
        for name, Us in side.items():
            row[name] = dict(objective=[float((W * (U @ U.t())).sum()) for U in Us], loss=[admission.perm_loss_of(W, U, sizes) for U in Us],
                             distinct_answers=len({(U @ U.t()).numpy().tobytes() for U in Us}))
        row["answers_shared_by_both_sides"] = len({(U @ U.t()).numpy().tobytes() for U in side["oracle"]} & {(U @ U.t()).numpy().tobytes() for U in side["device"]})
        for q in ("objective", "loss"):
            U, mu, var = mann_whitney(np.array(row["device"][q]), np.array(row["oracle"][q]))
            row["z_" + q] = (U - mu) / var ** 0.5 if var > 0 else 0.0
            pooled[q][0] += U - mu
            pooled[q][1] += var
        rows.append(row)
        print("input %d %s: objective oracle %.2f +- %.2f device %.2f +- %.2f (z %+.2f) | loss oracle %.5f device %.5f (z %+.2f) | distinct answers %d / %d, shared %d"
              % (j, sizes, np.mean(row["oracle"]["objective"]), np.std(row["oracle"]["objective"]), np.mean(row["device"]["objective"]),
                 np.std(row["device"]["objective"]), row["z_objective"], np.mean(row["oracle"]["loss"]), np.mean(row["device"]["loss"]), row["z_loss"],
                 row["oracle"]["distinct_answers"], row["device"]["distinct_answers"], row["answers_shared_by_both_sides"]), flush=True)
    out = dict(N=N, eps=1e-7, inputs=len(rows), pooled_z_objective=pooled["objective"][0] / pooled["objective"][1] ** 0.5,
               pooled_z_loss=pooled["loss"][0] / pooled["loss"][1] ** 0.5, seconds=time.time() - t0, rows=rows,
               reading="device minus oracle: positive z_objective = the device's objective <W, U U^T> tends to be HIGHER (better: the solver maximises it); "
                       "positive z_loss = higher matching loss.  |pooled z| <= 3 is what an exchangeable implementation gives 99.7 % of the time.")
    print("pooled over %d inputs x %d + %d draws: z objective %+.2f, z loss %+.2f" % (len(rows), N, N, out["pooled_z_objective"], out["pooled_z_loss"]))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
