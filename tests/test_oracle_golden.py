"""Pin the CPU oracle (oracle/gmodule.py) against golden vectors produced by the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import cases
from oracle import gmodule as og
from ttdg_mgm_amd import synth

torch.set_num_threads(1)
TOL = 1e-6  # restatement verified <=1e-6 against the imported reference


def close(a, b, tol=TOL):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert a.shape == tuple(b.shape), (a.shape, b.shape)
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    assert err <= tol, err


def check_pgrad(gold, key, g, tol=TOL):
    flat = g.detach().reshape(-1)
    close(flat[::cases.PSTRIDE], gold[key + "__sample"], tol)
    assert abs(float(flat.double().norm()) - float(gold[key + "__norm"])) <= tol * max(1.0, float(gold[key + "__norm"])) * 10


@pytest.mark.parametrize("ci", range(len(cases.AFF_CASES)))
def test_affinity(golden, ci):
    gold = golden("affinity")
    p = {k: v.clone().requires_grad_() for k, v in synth.mgm3_params(cases.AFF_PARAM_SEED).items()}
    X, Y, R = cases.aff_inputs(ci)
    X.requires_grad_(), Y.requires_grad_()
    M = og.affinity(p, X, Y)
    (M * R).sum().backward()
    close(M, gold[f"c{ci}_M"])
    close(X.grad, gold[f"c{ci}_dX"], 1e-5)
    close(Y.grad, gold[f"c{ci}_dY"], 1e-5)
    for k in ("fc_M.0.weight", "fc_M.0.bias", "fc_M.2.weight", "fc_M.2.bias", "project_sr.weight", "project_tg.weight"):
        check_pgrad(gold, f"c{ci}_d_{k}", p["node_affinity." + k].grad, 1e-4)


@pytest.mark.parametrize("ci", range(len(cases.MHA_CASES)))
def test_mha_adjacency(golden, ci):
    gold = golden("mha")
    p = synth.mgm3_params(cases.MHA_PARAM_SEED)
    close(og.mha_adjacency(p, cases.mha_input(ci)), gold[f"c{ci}_adj"])


@pytest.mark.parametrize("ci", range(len(cases.HUNG_CASES)))
def test_hungarian(golden, ci):
    gold = golden("hungarian")
    close(og.hungarian(cases.hung_input(ci)), gold[f"c{ci}_x"], 0)


def test_hungarian_ties(golden):
    gold = golden("hungarian")
    close(og.hungarian(torch.from_numpy(gold["ties_s"])), gold["ties_x"], 0)


def test_hungarian_bad_rank():
    with pytest.raises(ValueError):
        og.hungarian(torch.zeros(3))


@pytest.mark.parametrize("ci", range(3))
def test_permutation_loss(golden, ci):
    gold = golden("loss")
    s = torch.from_numpy(gold[f"c{ci}_s"]).requires_grad_()
    t = torch.from_numpy(gold[f"c{ci}_t"])
    l = og.permutation_loss(s.unsqueeze(0), t.unsqueeze(0))
    l.backward()
    close(l, gold[f"c{ci}_loss"])
    close(s.grad, gold[f"c{ci}_ds"])


def test_permutation_loss_range_assert():
    with pytest.raises(AssertionError):
        og.permutation_loss(torch.full((1, 2, 2), 1.5), torch.zeros(1, 2, 2))


@pytest.mark.parametrize("name,sizes,seed", cases.GAGM_CASES)
def test_gagm(golden, name, sizes, seed):
    gold = golden("gagm")
    A, W, U0 = cases.gagm_inputs(sizes, seed)
    tr = {}
    U = og.gagm(A, W, U0.clone(), sizes, trace=tr)
    close(tr["V0"], gold[f"{name}_V0"], 1e-5)
    close(U, gold[f"{name}_U"], 0)       # identical permutations
    assert tr["stages"][-1][0] == "hungarian" and len(tr["stages"]) == 6


@pytest.mark.parametrize("name", [c[0] for c in cases.MGM_CASES])
def test_mgm3_forward_backward(golden, name):
    gold = golden("mgm3")
    params, nodes, labels, U, sizes = cases.mgm_inputs(name)
    p = {k: v.clone().requires_grad_() for k, v in params.items()}
    nodes = [x.requires_grad_() for x in nodes]
    loss = og.mgm3_unsup_forward(p, nodes, labels, U)
    loss.backward()
    close(loss, gold[f"{name}_loss"])
    for gi, x in enumerate(nodes):
        close(x.grad, gold[f"{name}_dnode{gi}"])
    for k, v in p.items():
        if f"{name}_nograd_{k}" in gold:
            assert v.grad is None, k
        else:
            check_pgrad(gold, f"{name}_d_{k}", v.grad, 1e-5)


def test_mgm3_none_cases():
    nodes, labels = synth.node_sets(1, (9,))
    assert og.mgm3_unsup_forward(synth.mgm3_params(1), nodes, labels, synth.universe(2)) is None
    assert og.mgm3_unsup_forward(synth.mgm3_params(1), None, None, synth.universe(2)) is None


@pytest.mark.parametrize("ci", range(len(cases.PROTO_CASES)))
def test_prototype_computation(golden, ci):
    gold = golden("proto")
    name, feats, boxes, classes = cases.proto_inputs(ci)
    nodes, labels = og.prototype_computation(feats, boxes, classes)
    if f"{name}_none" in gold:
        assert nodes is None and labels is None
        return
    assert [len(n) for n in nodes] == gold[f"{name}_count"].tolist()
    for gi, (n, l) in enumerate(zip(nodes, labels)):
        close(n, gold[f"{name}_nodes{gi}"], 0)
        close(l, gold[f"{name}_labels{gi}"], 0)


@pytest.mark.parametrize("name", [c[0] for c in cases.PLANTED_CASES + cases.PLANTED_BIG_CASES])
def test_mgm3_planted_cases(golden, name):
    """Trained-like planted cases: the reference converges and is rounding-stable, so U itself is golden."""
    gold = golden("mgm3_big" if name.startswith("pb_") else "mgm3")
    params, nodes, labels, U, sizes = cases.mgm_inputs(name)
    p = {k: v.clone().requires_grad_() for k, v in params.items()}
    nodes = [x.requires_grad_() for x in nodes]
    tr = {}
    loss = og.mgm3_unsup_forward(p, nodes, labels, U, trace=tr)
    loss.backward()
    close(loss, gold[f"{name}_loss"])
    close(tr["Ub"], gold[f"{name}_U"], 0)
    assert max(tr["iters"]) < 200
    for gi, x in enumerate(nodes):
        close(x.grad, gold[f"{name}_dnode{gi}"], 1e-5)


@pytest.mark.parametrize("name", [c[0] for c in cases.PLANTED_CASES + cases.PLANTED_BIG_CASES])
def test_planted_goldens_do_not_sit_on_a_rounding_edge(golden, name):
    """VERDICT r2 item 1b.  The admission test make_golden.py applies with the reference's solver, repeated with the oracle
    against the committed golden U: a float64 front end, reversed summation order, per-block +-2 ulp on Wds, the solve in
    float64, and 1e-6 / 1e-5 relative noise in EVERY Sinkhorn-stage projection all return the golden permutation."""
    import admission
    gold = golden("mgm3_big" if name.startswith("pb_") else "mgm3")
    params, nodes, labels, U, sizes = cases.mgm_inputs(name)
    ok, res = admission.check(params, nodes, labels, U, sizes, golden=torch.from_numpy(gold[f"{name}_U"]))
    assert ok, res


# ----------------------------------------------------------------------------- N3: HiPPI / U_sup
@pytest.mark.parametrize("name,sizes,seed,proj", cases.HIPPI_CASES)
def test_hippi(golden, name, sizes, seed, proj):
    gold = golden("usup")
    W, U0 = cases.hippi_inputs(sizes, seed)
    tr = {}
    U = og.hippi(W, U0, sizes, 32, projector=proj, trace=tr)
    close(tr["V0"], gold[f"hippi_{name}_V0"], 1e-5)
    close(U, gold[f"hippi_{name}_U"], 1e-5)


def test_hippi_bad_projector():
    W, U0 = cases.hippi_inputs((5, 6), 1)
    with pytest.raises(NameError):
        og.hippi(W, U0, (5, 6), 32, projector="nope")


@pytest.mark.parametrize("name,sizes,seed", cases.USUP_CASES)
def test_u_sup_forward_backward(golden, name, sizes, seed):
    """Everything that carries gradient in U_sup.forward (:136-158), with the (detached) HiPPI target taken from the
    reference run; the free-running target is rounding-driven (edges ~1e7) and only has to be finite."""
    gold = golden("usup")
    p = {k: v.clone().requires_grad_() for k, v in synth.usup_params(cases.USUP_PARAM_SEED).items()}
    nodes, labels = cases.usup_inputs(sizes, seed)
    nodes = [x.requires_grad_() for x in nodes]
    tr = {}
    loss = og.u_sup_forward(p, nodes, labels, forced_target=torch.from_numpy(gold[f"usup_{name}_target"]), trace=tr)
    loss.backward()
    close(tr["N"], gold[f"usup_{name}_N"], 1e-5)
    close(tr["Us"], gold[f"usup_{name}_Us"], 1e-6)
    close(loss, gold[f"usup_{name}_loss"], 1e-7)
    for g, x in enumerate(nodes):
        close(x.grad, gold[f"usup_{name}_dnode{g}"], 1e-7)
    close(p["U"].grad, gold[f"usup_{name}_d_U"], 1e-7)
    for k in ("linear_k.weight", "linear_v.weight", "linear_q.weight", "linear_final.weight", "linear_final.bias", "layer_norm.weight"):
        check_pgrad(gold, f"usup_{name}_d_Net_U.g_gene.{k}", p["Net_U.g_gene." + k].grad, 1e-6)
    free = og.u_sup_forward(p, [x.detach() for x in nodes], labels)
    assert torch.isfinite(free)


# ---- trained-regime solver inputs (A / Wds / U0 recorded by tools/gagm_trained_probe.py on the synthetic checkpoint) ------
@pytest.mark.parametrize("j", range(4))
def test_gagm_trained_regime_is_reproducible_through_four_stages(golden, j):
    """On trained-regime inputs the solve is chaotic in its LAST Sinkhorn stage (tau = 0.00625: float32 and float64 runs of
    this very restatement end in different permutations), but the trajectory through the first four stages
    (tau 0.1 ... 0.0125, 15-50 free-running iterations) is not: that is the part the device is held to
    (tests/test_gpu_parity.py::test_gagm_trained_regime_trajectory)."""
    sizes, A, _, W, U0 = cases.trained_solver_case(golden("trained_solver_inputs"), j)
    t32, t64 = {}, {}
    U32 = og.gagm(A, W, U0, sizes, trace=t32, max_stages=4)
    U64 = og.gagm(A.double(), W.double(), U0.double(), sizes, trace=t64, max_stages=4)
    assert t32["iters"] == t64["iters"] and len(t32["iters"]) == 4
    assert float((U32.double() - U64).abs().max()) <= 5e-6


@pytest.mark.parametrize("name", [c[0] for c in cases.PLANTED_CFG3_CASES])
def test_cfg3_size_planted_golden(golden, name):
    """BASELINE.json cfg-3 at full size (8 x 256 nodes): the oracle against the REFERENCE's own run of the planted case
    (tests/golden/mgm3_cfg3.npz; multi_graph_matching.py:487-569 + :300-389): Wds (strided sample + norm), U0, first V, the
    iteration count of every stage as the reference's print_helper reported it, the permutation matrices, and the loss."""
    gold = golden("mgm3_cfg3")
    params, nodes, labels, U, sizes = cases.mgm_inputs(name)
    tr = {}
    with torch.no_grad():
        loss = og.mgm3_unsup_forward(params, nodes, labels, U, trace=tr)
    W = tr["Wds"].reshape(-1)
    close(W[::cases.CFG3_WSTRIDE], gold[f"{name}_Wds__sample"], 1e-6)
    assert abs(float(W.double().norm()) - float(gold[f"{name}_Wds__norm"])) <= 1e-5 * float(gold[f"{name}_Wds__norm"])
    close(tr["U0"], gold[f"{name}_U0"], 1e-6)
    close(tr["V0"], gold[f"{name}_V0"], 1e-5)
    assert tr["iters"] == gold[f"{name}_iters"].tolist()
    assert np.array_equal(cases.perm_to_columns(tr["Ub"].numpy()), gold[f"{name}_U"])
    assert np.array_equal(cases.columns_to_perm(gold[f"{name}_U"]), tr["Ub"].numpy())
    close(loss, gold[f"{name}_loss"], 1e-6)
