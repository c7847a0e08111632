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

Two perturbation sizes: 1e-7 (one ulp: the round-4 census' size) and 1e-5.  The two implementations' own arithmetic differs by ~1e-6
(max |Wds_device - Wds_oracle|), so under 1e-7 noise each side samples a ball SMALLER than the distance between the two
sides' states - where the basin structure of a chaotic map has features at that scale the two balls see different mixtures (recorded:
an input on which the oracle returns ONE answer for 64 draws and the device 25) - while 1e-5 noise makes both sides sample the same
neighbourhood: that is the size at which exchangeability is the right null hypothesis.

usage:  python tools/census_exchangeability.py oracle [N=48]      (CPU: the oracle side only -> tools/fixtures/census_exchangeability_oracle.json)
        python tools/census_exchangeability.py [N=48] [out.json]  (GPU box: the device side; the oracle side from the fixture when its N matches)"""
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


def perturbed(W, U0, seed, eps):
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


EPS = (1e-7, 1e-5)
ORACLE_FIXTURE = os.path.join(ROOT, "tools", "fixtures", "census_exchangeability_oracle.json")


def answer_key(U):
    import hashlib
    return hashlib.sha1((U @ U.t()).numpy().tobytes()).hexdigest()[:16]


def summarise(W, Us, sizes):
    import admission
    return dict(objective=[float((W * (U @ U.t())).sum()) for U in Us], loss=[admission.perm_loss_of(W, U, sizes) for U in Us],
                answers=[answer_key(U) for U in Us])


def oracle_side(N):
    import admission
    from oracle import gmodule as og
    inputs = torch.load(os.path.join(ROOT, "tools", "fixtures", "trained_solver_inputs.pt"), weights_only=True)
    out = dict(N=N, eps=list(EPS), rows=[])
    for j, rec in enumerate(inputs):
        sizes = [int(n) for n in rec["sizes"]]
        apack, W, U0 = rec["apack"].float(), rec["Wds"].float(), rec["U0"].float()
        A = admission.unpack_adjacency(apack, sizes)
        row = {}
        for ei, eps in enumerate(EPS):
            Us = [og.gagm(A, *perturbed(W, U0, 5000 + 10000 * ei + 100 * j + k, eps), sizes) for k in range(N)]
            row["%g" % eps] = summarise(W, Us, sizes)
        out["rows"].append(row)
        print("oracle side: input %d done" % j, flush=True)
    return out


def main():
    torch.set_num_threads(8)          # 100-node problems: more threads only add synchronisation
    if len(sys.argv) > 1 and sys.argv[1] == "oracle":
        N = int(sys.argv[2]) if len(sys.argv) > 2 else 48
        with open(ORACLE_FIXTURE, "w") as f:
            json.dump(oracle_side(N), f)
        return
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "census_exchangeability.json")
    from ttdg_mgm_amd import ops
    dev = torch.device("cuda:0")
    inputs = torch.load(os.path.join(ROOT, "tools", "fixtures", "trained_solver_inputs.pt"), weights_only=True)
    orc = None
    if os.path.exists(ORACLE_FIXTURE):
        with open(ORACLE_FIXTURE) as f:
            orc = json.load(f)
        if orc["N"] != N or orc["eps"] != list(EPS):
            orc = None
    if orc is None:
        orc = oracle_side(N)
    t0 = time.time()
    out = dict(N=N, inputs=len(inputs), by_eps={})
    for ei, eps in enumerate(EPS):
        rows, pooled = [], {"objective": [0.0, 0.0], "loss": [0.0, 0.0]}
        for j, rec in enumerate(inputs):
            sizes = [int(n) for n in rec["sizes"]]
            apack, W, U0 = rec["apack"].float(), rec["Wds"].float(), rec["U0"].float()
            gr = ops.graphs(sizes)
            Us = []
            for k in range(N):
                Wp, Up = perturbed(W, U0, 7000 + 10000 * ei + 100 * j + k, eps)
                Us.append(ops.gagm_solve(apack.to(dev), Wp.to(dev), Up.to(dev), gr, sizes)[0].cpu())
            row = dict(input=j, sizes=sizes, oracle=orc["rows"][j]["%g" % eps], device=summarise(W, Us, sizes))
            for name in ("oracle", "device"):
                row[name]["distinct_answers"] = len(set(row[name]["answers"]))
            row["answers_shared_by_both_sides"] = len(set(row["oracle"]["answers"]) & set(row["device"]["answers"]))
            for q in ("objective", "loss"):
                U, mu, var = mann_whitney(np.array(row["device"][q]), np.array(row["oracle"][q]))
                row["z_" + q] = (U - mu) / var ** 0.5 if var > 0 else 0.0
                pooled[q][0] += U - mu
                pooled[q][1] += var
            rows.append(row)
            print("eps %g input %d %s: objective oracle %.2f +- %.2f device %.2f +- %.2f (z %+.2f) | loss oracle %.5f device %.5f (z %+.2f) | distinct answers %d / %d, shared %d"
                  % (eps, j, sizes, np.mean(row["oracle"]["objective"]), np.std(row["oracle"]["objective"]), np.mean(row["device"]["objective"]),
                     np.std(row["device"]["objective"]), row["z_objective"], np.mean(row["oracle"]["loss"]), np.mean(row["device"]["loss"]), row["z_loss"],
                     row["oracle"]["distinct_answers"], row["device"]["distinct_answers"], row["answers_shared_by_both_sides"]), flush=True)
        zo = pooled["objective"][0] / max(pooled["objective"][1], 1e-30) ** 0.5
        zl = pooled["loss"][0] / max(pooled["loss"][1], 1e-30) ** 0.5
        out["by_eps"]["%g" % eps] = dict(pooled_z_objective=zo, pooled_z_loss=zl, per_input_z_objective=[r["z_objective"] for r in rows],
                                         per_input_z_loss=[r["z_loss"] for r in rows], rows=rows)
        print("eps %g pooled over %d inputs x %d + %d draws: z objective %+.2f, z loss %+.2f" % (eps, len(rows), N, N, zo, zl), flush=True)
    out["seconds"] = time.time() - t0
    out["reading"] = ("device minus oracle: positive z_objective = the device's objective <W, U U^T> tends to be HIGHER (better: the solver maximises it); "
                      "positive z_loss = higher matching loss.  |pooled z| <= 3 is what an exchangeable implementation gives 99.7 % of the time.")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
