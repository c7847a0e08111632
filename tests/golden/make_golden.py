"""Generate golden vectors by running the REFERENCE's own modules (build container only).

    python tests/golden/make_golden.py

Imports /root/reference read-only through oracle/ref_import.py (namespace
stubs + our Sinkhorn spec standing in for the absent pygmtools), feeds it the
repo-owned PCG64 inputs of ttdg_mgm_amd.synth, and writes inputs-by-seed +
reference outputs to tests/golden/*.npz.  Only data is written: no reference
source text in any encoding.  MHA is put in .eval() (dropout off) so the
reference is repeatable (SURVEY.md §7 'Dropout in train mode').
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from ttdg_mgm_amd import synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cases import *  # noqa: E402,F401,F403
import admission  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)  # fixed reduction order for the goldens


def ref_mgm3(mgm, seed):
    m = mgm.MGM3_unsup(2, 32)
    m.load_state_dict(synth.mgm3_params(seed), strict=True)
    m.eval()
    return m


def npy(t):
    return t.detach().cpu().numpy()


# parameter gradients are stored as a strided sample (cases.PSTRIDE) + L2 norm: keeps fixtures small


def pgrad(out, key, g):
    flat = g.detach().reshape(-1)
    out[key + "__sample"] = npy(flat[::PSTRIDE])
    out[key + "__norm"] = npy(flat.double().norm())


def gold_affinity(mgm):
    out = {}
    m = ref_mgm3(mgm, AFF_PARAM_SEED)
    for ci, (n1, n2) in enumerate(AFF_CASES):
        X, Y, R = aff_inputs(ci)
        X.requires_grad_(), Y.requires_grad_()
        m.zero_grad()
        M = m.node_affinity(X, Y)
        (M * R).sum().backward()
        out[f"c{ci}_M"] = npy(M)
        out[f"c{ci}_dX"] = npy(X.grad)
        out[f"c{ci}_dY"] = npy(Y.grad)
        for k, p in m.node_affinity.named_parameters():
            pgrad(out, f"c{ci}_d_{k}", p.grad)
    np.savez_compressed(os.path.join(OUT, "affinity.npz"), **out)


def gold_mha(mgm):
    out = {}
    m = ref_mgm3(mgm, MHA_PARAM_SEED)
    for ci, n in enumerate(MHA_CASES):
        x = mha_input(ci)
        _, adj = m.intra_domain_graph([x, x, x])
        out[f"c{ci}_adj"] = npy(adj)
    np.savez_compressed(os.path.join(OUT, "mha.npz"), **out)


def gold_hungarian(mgm):
    out = {}
    for ci, (r, c) in enumerate(HUNG_CASES):
        s = hung_input(ci)
        out[f"c{ci}_x"] = npy(mgm.hungarian(s))
    # structured ties: empty universe columns give exact-zero columns in V
    g = synth.gen(320)
    s = synth.normal(g, (12, 32), 1.0).abs()
    s[:, [3, 7, 8, 20, 21, 22, 30]] = 0.0
    s[4:9, :] = 0.0
    out["ties_s"] = npy(s)
    out["ties_x"] = npy(mgm.hungarian(s))
    np.savez_compressed(os.path.join(OUT, "hungarian.npz"), **out)


def gold_loss(mgm):
    out = {}
    crit = mgm.PermutationLoss()
    for ci, (r, c) in enumerate(((5, 9), (22, 22), (40, 33))):
        g = synth.gen(400 + ci)
        s = torch.from_numpy(g.uniform(0, 1, size=(r, c)).astype(np.float32))
        s[0, 0], s[-1, -1] = 0.0, 1.0  # exercise the clamp
        s.requires_grad_()
        t = torch.from_numpy((g.uniform(0, 1, size=(r, c)) < 0.1).astype(np.float32))
        l = crit(s.unsqueeze(0), t.unsqueeze(0), torch.tensor(r), torch.tensor(c))
        l.backward()
        out[f"c{ci}_s"], out[f"c{ci}_t"] = npy(s), npy(t)
        out[f"c{ci}_loss"], out[f"c{ci}_ds"] = npy(l), npy(s.grad)
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)


def gold_gagm(mgm):
    out = {}
    solver = mgm.GA_GM(mgm_iter=[200], cluster_iter=10, sk_iter=20, sk_tau0=[0.1], sk_gamma=0.5,
                       cluster_beta=[1.0, 0.0], converge_tol=1.0e-3, min_tau=[1.0e-2],
                       projector0=["sinkhorn", "sinkhorn"])
    for name, sizes, seed in GAGM_CASES:
        A, W, U0 = gagm_inputs(sizes, seed)
        ms = torch.tensor(sizes, dtype=torch.int)
        Ub, cluster = solver(A, W, U0.clone(), ms, 32, 0.5, 1)
        out[f"{name}_U"] = npy(Ub)
        # first-iteration V exactly as multi_graph_matching.py:317-321 computes it
        UUt = U0 @ U0.t()
        V = torch.chain_matmul(A, UUt, A, U0) * 0.5 * 2 + W @ U0
        out[f"{name}_V0"] = npy(V / len(sizes))
    np.savez_compressed(os.path.join(OUT, "gagm.npz"), **out)


def ref_solve(mgm, A, W, U0, sizes, perturb=None):
    """The reference's own GA_GM (multi_graph_matching.py:223-244,300-389) on given solver inputs.  ``perturb`` multiplies
    rounding-sized noise into every projection of the Sinkhorn stages by wrapping the reference's Sinkhorn.forward_log
    (utils/sinkhorn.py:85-87) for the duration of the call."""
    solver = mgm.GA_GM(mgm_iter=[200], cluster_iter=10, sk_iter=20, sk_tau0=[0.1], sk_gamma=0.5, cluster_beta=[1.0, 0.0],
                       converge_tol=1.0e-3, min_tau=[1.0e-2], projector0=["sinkhorn", "sinkhorn"])
    orig = mgm.Sinkhorn.forward_log
    if perturb is not None:
        def noisy(self, *a, **k):
            out = orig(self, *a, **k)
            return perturb(out) if self.batched_operation else out
        mgm.Sinkhorn.forward_log = noisy
    try:
        Ub, _ = solver(A, W, U0.clone(), torch.tensor(list(sizes), dtype=torch.int), 32, 0.5, 1)
    finally:
        mgm.Sinkhorn.forward_log = orig
    return Ub.detach()


def planted_goldens(mgm, case_list, out):
    """Free-running reference runs whose permutation matrices are golden: asserted rounding-stable (1 thread, 8 threads,
    two 1e-7-relative input perturbations give the same U)."""
    for name, sizes, seed in case_list:
        params, nodes, labels, U, _ = mgm_inputs(name)
        runs = []
        for trial in range(4):          # 1 thread, 8 threads, and two 1e-7-relative input perturbations
            torch.set_num_threads(8 if trial == 1 else 1)
            m = mgm.MGM3_unsup(2, 32)
            m.load_state_dict(params, strict=True)
            m.eval()
            xs = [x.clone() for x in nodes]
            if trial >= 2:
                g = synth.gen(77 + trial)
                xs = [x * (1 + 1e-7 * synth.normal(g, tuple(x.shape))) for x in xs]
            xs = [x.requires_grad_() for x in xs]
            cap = {}
            orig = m.ga_mgmc.forward

            def spy(*a, _o=orig, _c=cap, **k):
                out = _o(*a, **k)
                _c["U"] = out[0].detach().clone()
                return out
            m.ga_mgmc.forward = spy
            loss = m(xs, labels, U)
            loss.backward()
            runs.append((cap["U"], loss.detach(), m, xs))
        torch.set_num_threads(1)
        assert all(torch.equal(runs[0][0], r[0]) for r in runs[1:]), "reference not rounding-stable on " + name
        Ub, loss, m, xs = runs[0]
        # admission (tests/golden/admission.py): the REFERENCE's GA_GM must return this same U under structured
        # rounding-sized perturbations of its inputs and of every Sinkhorn-stage projection
        ok, res = admission.check(params, nodes, labels, U, sizes, solve=lambda *a, **k: ref_solve(mgm, *a, **k), golden=Ub)
        print("admission %-9s %s" % (name, "ok" if ok else "FAILED: " + ", ".join(k for k, v in res.items() if not v)), flush=True)
        assert ok, "planted case %s sits on a rounding edge of the reference: %s" % (name, res)
        out[f"{name}_loss"] = npy(loss)
        out[f"{name}_U"] = npy(Ub)
        for gi, x in enumerate(xs):
            out[f"{name}_dnode{gi}"] = npy(x.grad)
        for k, p in m.named_parameters():
            if p.grad is not None:
                pgrad(out, f"{name}_d_{k}", p.grad)
            else:
                out[f"{name}_nograd_{k}"] = np.zeros(0, np.float32)


def gold_mgm3_big(mgm):
    """Planted cases where the kernels branch: n_g > 32, 64 < n_g <= 128, the multi-workgroup solver, G = 2 with n > 32."""
    out = {}
    planted_goldens(mgm, PLANTED_BIG_CASES, out)
    np.savez_compressed(os.path.join(OUT, "mgm3_big.npz"), **out)


def gold_mgm3_cfg3(mgm):
    """BASELINE.json cfg-3 at full size (8 graphs x 256 nodes): the REFERENCE's own free-running
    MGM3_unsup.forward + backward on the planted case cases.PLANTED_CFG3_CASES, admitted like every planted golden (1 / 8
    threads, two 1e-7 input perturbations, tests/golden/admission.py with the reference's GA_GM), stored compactly:
      U            one uint8 universe column per node (255 = unassigned)                       multi_graph_matching.py:300-389
      iters        iterations per stage of the schedule, read off the reference's own print_helper(i, tau) calls  :374-381
      V0           first-iteration V exactly as :317-321 computes it
      U0           x U^T (:531-532);  Wds as a strided sample (cases.CFG3_WSTRIDE) + Frobenius norm (:518-525)
      loss, node / parameter gradients (strided sample + L2 norm)."""
    out = {}
    for name, sizes, seed, _ in PLANTED_CFG3_CASES:
        params, nodes, labels, U, _ = mgm_inputs(name)
        runs = []
        for trial in range(4):
            torch.set_num_threads(8 if trial == 1 else 1)
            m = mgm.MGM3_unsup(2, 32)
            m.load_state_dict(params, strict=True)
            m.eval()
            xs = [x.clone() for x in nodes]
            if trial >= 2:
                g = synth.gen(77 + trial)
                xs = [x * (1 + 1e-7 * synth.normal(g, tuple(x.shape))) for x in xs]
            xs = [x.requires_grad_() for x in xs]
            cap = {"stages": []}
            orig = m.ga_mgmc.forward

            def spy(A, W, U0, *a, _o=orig, _c=cap, **k):
                _c["A"], _c["W"], _c["U0"] = A.detach().clone(), W.detach().clone(), U0.detach().clone()
                res = _o(A, W, U0, *a, **k)
                _c["U"] = res[0].detach().clone()
                return res
            m.ga_mgmc.forward = spy
            ph = mgm.print_helper
            mgm.print_helper = lambda i, tag, _c=cap: _c["stages"].append((int(i) + 1, tag))
            try:
                loss = m(xs, labels, U)
            finally:
                mgm.print_helper = ph
            if trial == 0:
                loss.backward()
            runs.append((cap, loss.detach(), m, xs))
            print("cfg3 golden %s trial %d: loss %.6f stages %s" % (name, trial, float(loss), cap["stages"]), flush=True)
        torch.set_num_threads(1)
        assert all(torch.equal(runs[0][0]["U"], r[0]["U"]) for r in runs[1:]), "reference not rounding-stable on " + name
        assert all([c for c, _ in runs[0][0]["stages"]] == [c for c, _ in r[0]["stages"]] for r in runs[1:]), "stage counts move under rounding on " + name
        cap, loss, m, xs = runs[0]
        Ub = cap["U"]
        torch.set_num_threads(8)
        ok, res = admission.check(params, nodes, labels, U, sizes, solve=lambda *a, **k: ref_solve(mgm, *a, **k), golden=Ub)
        torch.set_num_threads(1)
        print("admission %-9s %s" % (name, "ok" if ok else "FAILED: " + ", ".join(k for k, v in res.items() if not v)), flush=True)
        assert ok, "planted case %s sits on a rounding edge of the reference: %s" % (name, res)
        A, W, U0 = cap["A"], cap["W"], cap["U0"]
        V0 = (torch.chain_matmul(A, U0 @ U0.t(), A, U0) * 0.5 * 2 + W @ U0) / len(sizes)
        out[f"{name}_U"] = perm_to_columns(npy(Ub))
        out[f"{name}_iters"] = np.array([c for c, _ in cap["stages"]], np.int64)
        out[f"{name}_V0"] = npy(V0)
        out[f"{name}_U0"] = npy(U0)
        out[f"{name}_Wds__sample"] = npy(W.reshape(-1)[::CFG3_WSTRIDE])
        out[f"{name}_Wds__norm"] = npy(W.double().norm())
        out[f"{name}_loss"] = npy(loss)
        for gi, x in enumerate(xs):
            pgrad(out, f"{name}_dnode{gi}", x.grad)
        for k, p in m.named_parameters():
            if p.grad is not None:
                pgrad(out, f"{name}_d_{k}", p.grad)
            else:
                out[f"{name}_nograd_{k}"] = np.zeros(0, np.float32)
    np.savez_compressed(os.path.join(OUT, "mgm3_cfg3.npz"), **out)


def gold_mgm3(mgm):
    out = {}
    for name, sizes, seed in MGM_CASES:
        params, nodes, labels, U, _ = mgm_inputs(name)
        m = mgm.MGM3_unsup(2, 32)
        m.load_state_dict(params, strict=True)
        m.eval()
        nodes = [x.requires_grad_() for x in nodes]
        m.zero_grad()
        cap = {}
        orig = m.ga_mgmc.forward

        def spy(*a, _o=orig, _c=cap, **k):
            res = _o(*a, **k)
            _c["U"] = res[0].detach().clone()
            return res
        m.ga_mgmc.forward = spy
        loss = m(nodes, labels, U)
        loss.backward()
        out[f"{name}_loss"] = npy(loss)
        out[f"{name}_U"] = npy(cap["U"])      # the pseudo-labels this (rounding-unstable) run happened to produce
        for gi, x in enumerate(nodes):
            out[f"{name}_dnode{gi}"] = npy(x.grad)
        for k, p in m.named_parameters():
            if p.grad is not None:
                pgrad(out, f"{name}_d_{k}", p.grad)
            else:
                out[f"{name}_nograd_{k}"] = np.zeros(0, np.float32)
    # planted (trained-like) cases: the solver converges, the reference is rounding-stable -> U is golden too
    planted_goldens(mgm, PLANTED_CASES, out)
    # single graph -> None (multi_graph_matching.py:489-490)
    m = ref_mgm3(mgm, 1)
    nodes, labels = synth.node_sets(1, (9,))
    assert m(nodes, labels, synth.universe(2)) is None
    assert m(None, None, synth.universe(2)) is None
    np.savez_compressed(os.path.join(OUT, "mgm3.npz"), **out)


def gold_proto(bg):
    out = {}
    pc = bg.PrototypeComputation(2, 10)
    for ci, (name, size, per_img) in enumerate(PROTO_CASES):
        _, feats, boxes, classes = proto_inputs(ci)
        insts = [ref_import.FakeInstances(b, c) for b, c in zip(boxes, classes)]
        nodes, labels = pc(feats, insts)
        if nodes is None:
            out[f"{name}_none"] = np.ones(1, np.int64)
            continue
        out[f"{name}_count"] = np.array([len(n) for n in nodes], np.int64)
        for gi, (n, l) in enumerate(zip(nodes, labels)):
            out[f"{name}_nodes{gi}"] = npy(n)
            out[f"{name}_labels{gi}"] = npy(l)
    np.savez_compressed(os.path.join(OUT, "proto.npz"), **out)


def gold_dice():
    """Dice / E-measure / S-measure of the reference's own numpy functions (evaluation/dice_metric.py:54-66,110-240)."""
    dm = ref_import.load_dice_metric()
    np.bool = bool     # the reference still uses the removed numpy alias; harmless here
    out = {}
    for i, (p, g) in enumerate(dice_mask_pairs()):
        inter = np.logical_and(p, g).sum()
        out[f"c{i}_dice"] = np.float64(2 * inter / (p.sum() + g.sum() + 1e-6))
        out[f"c{i}_ea"] = np.float64(dm.enhanced_align(p, g))
        out[f"c{i}_sm"] = np.float64(dm.Structure_measure().get_score(p, g))
    np.savez_compressed(os.path.join(OUT, "dice.npz"), **out)


def gold_sinkhorn_ref():
    """Outputs of the reference tree's own log-Sinkhorn (graph_matching.py:828-839, slack=False) in the LOG domain."""
    fn = ref_import.load_tree_sinkhorn()
    out = {}
    for name, b, r, c, tau, seed, scale in SKREF_CASES:
        out[f"{name}_log"] = npy(fn(skref_log_alpha(name), SKREF_SWEEPS, slack=False))
    np.savez_compressed(os.path.join(OUT, "sinkhorn_ref.npz"), **out)


def gold_usup(mgm):
    """N3: HiPPI on planted similarities, and U_sup.forward with the HiPPI result captured so that everything carrying
    gradient can be pinned independently of the rounding-driven edge weights."""
    out = {}
    solver = mgm.HiPPI()
    for name, sizes, seed, proj in HIPPI_CASES:
        W, U0 = hippi_inputs(sizes, seed)
        out[f"hippi_{name}_U"] = npy(solver(W, U0, torch.tensor(sizes), 32, projector=proj))
        WU = W @ U0
        out[f"hippi_{name}_V0"] = npy(torch.chain_matmul(WU, U0.t(), WU))
    for name, sizes, seed in USUP_CASES:
        m = mgm.U_sup(2, 32)
        m.load_state_dict(synth.usup_params(USUP_PARAM_SEED), strict=True)
        m.eval()
        nodes, labels = usup_inputs(sizes, seed)
        nodes = [x.requires_grad_() for x in nodes]
        cap = {}
        net, match, sk = m.Net_U.forward, m.matching.forward, m.sinkhorn.forward

        def spy_net(*a, _o=net, **k):
            N, E = _o(*a, **k)
            cap["N"], cap["E"] = N.detach().clone(), [e.detach().clone() for e in E]
            return N, E

        def spy_match(W, U, ms, d, _o=match, **k):
            cap["A_"], cap["Us"] = W.detach().clone(), U.detach().clone()
            cap["target"] = _o(W, U, ms, d, **k).detach().clone()
            return cap["target"]

        m.Net_U.forward, m.matching.forward = spy_net, spy_match
        loss = m(nodes, labels)
        loss.backward()
        out[f"usup_{name}_N"] = npy(cap["N"])
        out[f"usup_{name}_Us"] = npy(cap["Us"])
        out[f"usup_{name}_target"] = npy(cap["target"])
        out[f"usup_{name}_loss"] = npy(loss)
        out[f"usup_{name}_edge_absmax"] = npy(torch.stack([e.abs().max() for e in cap["E"]]))
        for g, x in enumerate(nodes):
            out[f"usup_{name}_dnode{g}"] = npy(x.grad)
        for k, p in m.named_parameters():
            if p.grad is not None:
                if k == "U":
                    out[f"usup_{name}_d_U"] = npy(p.grad)
                else:
                    pgrad(out, f"usup_{name}_d_{k}", p.grad)
    np.savez_compressed(os.path.join(OUT, "usup.npz"), **out)


def main():
    mgm, bg = ref_import.load()
    if len(sys.argv) > 1 and sys.argv[1] == "usup":
        gold_usup(mgm)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mgm3_big":
        gold_mgm3_big(mgm)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mgm3_cfg3":
        gold_mgm3_cfg3(mgm)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mgm3":
        gold_mgm3(mgm)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "sinkhorn_ref":
        gold_sinkhorn_ref()
        return
    gold_dice()
    gold_affinity(mgm)
    gold_mha(mgm)
    gold_hungarian(mgm)
    gold_loss(mgm)
    gold_gagm(mgm)
    gold_mgm3(mgm)
    gold_mgm3_big(mgm)
    gold_mgm3_cfg3(mgm)
    gold_proto(bg)
    gold_usup(mgm)
    gold_sinkhorn_ref()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
