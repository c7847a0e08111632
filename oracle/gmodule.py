"""torch-CPU fp32 restatement of the reference GModule hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py): the checker for the HIP path and
the timed "port" CPU baseline of bench.py.  Same formulation as the reference
on purpose — materialised N1xN2x512 affinity MLP, per-pair Python loops, one
scipy LAP per graph per Hungarian-stage iteration — so that timing it is
timing the reference's algorithm, not ours.

Every function cites the reference lines it follows (paths relative to
/root/reference/adapteacher/modeling/GModule/).  Pinned against the imported
reference modules by tests/golden/*.npz (tests/test_oracle_golden.py).

Parameters travel as a flat dict keyed by the reference's state-dict names
relative to the owning module, e.g. for MGM3_unsup:
    node_affinity.project_sr.weight, node_affinity.project_tg.weight,
    node_affinity.fc_M.0.weight, node_affinity.fc_M.0.bias,
    node_affinity.fc_M.2.weight, node_affinity.fc_M.2.bias,
    intra_domain_graph.linear_{k,v,q,final}.{weight,bias},
    intra_domain_graph.layer_norm.{weight,bias}
"""
import itertools

import numpy as np
import scipy.optimize
import torch
import torch.nn.functional as F

from .sinkhorn_spec import sinkhorn as _sinkhorn

# hard-coded matching hyper-parameters (multi_graph_matching.py:455-474, rcnn.py:115-116)
UNIV_SIZE = 32
SAMPLE_DIST = 10
PAIR_SK_ITER, PAIR_SK_TAU = 20, 0.05
GA_MGM_ITER, GA_SK_ITER, GA_TAU0, GA_GAMMA, GA_TOL, GA_MIN_TAU = 200, 20, 0.1, 0.5, 1.0e-3, 1.0e-2
QUAD_WEIGHT = 0.5
FOCAL_ALPHA, FOCAL_GAMMA, FOCAL_EPS = 0.25, 2, 1e-6


# --------------------------------------------------------------------------- A4
def affinity(p, X, Y, prefix="node_affinity."):
    """utils/affinity.py:44-57 — M_ij = fc2(relu(fc1([Psr x_i ; Ptg y_j])))."""
    Xp = F.linear(X, p[prefix + "project_sr.weight"])
    Yp = F.linear(Y, p[prefix + "project_tg.weight"])
    n1, n2, c = Xp.shape[0], Yp.shape[0], Xp.shape[1]
    grid = torch.cat([Xp.unsqueeze(1).expand(n1, n2, c), Yp.unsqueeze(0).expand(n1, n2, c)], dim=-1)
    h = torch.relu(F.linear(grid, p[prefix + "fc_M.0.weight"], p[prefix + "fc_M.0.bias"]))
    return F.linear(h, p[prefix + "fc_M.2.weight"], p[prefix + "fc_M.2.bias"]).squeeze()


# --------------------------------------------------------------------------- A3
def mha_adjacency(p, x, prefix="intra_domain_graph.", dropout_mask=None):
    """utils/attentions.py:60-86 (version 'v2', 1 head) — only ``attention`` is used
    by the caller (multi_graph_matching.py:498,571-574).  scale = 256**-0.5 (:80).
    ``dropout_mask`` (n,n) of 0/1 reproduces a train-mode draw (attentions.py:40);
    None = eval mode."""
    d = x.shape[1]
    k = F.linear(x, p[prefix + "linear_k.weight"], p[prefix + "linear_k.bias"])
    q = F.linear(x, p[prefix + "linear_q.weight"], p[prefix + "linear_q.bias"])
    att = torch.softmax((q @ k.t()) * (d ** -0.5), dim=1)
    if dropout_mask is not None:
        att = att * dropout_mask / 0.9
    return att


# --------------------------------------------------------------------------- A7
def hungarian(s):
    """utils/hungarian.py:8-66 — maximise sum(s) over assignments; 0/1 matrix of s's shape.
    2-D input only on this path; float32 -> scipy (float64 inside)."""
    if s.dim() != 2:
        raise ValueError("input data shape not understood: {}".format(s.shape))
    neg = s.detach().cpu().numpy() * -1
    rows, cols = scipy.optimize.linear_sum_assignment(neg)
    out = np.zeros_like(neg)
    out[rows, cols] = 1
    return torch.from_numpy(out).to(s.device)


def pad_tensor(ts):
    """utils/pad_tensor.py:5-31 — zero-pad to the common max shape."""
    shape = [max(t.shape[i] for t in ts) for i in range(ts[0].dim())]
    out = []
    for t in ts:
        pad = []
        for i in reversed(range(t.dim())):
            pad += [0, shape[i] - t.shape[i]]
        out.append(F.pad(t, pad, "constant", 0))
    return out


# --------------------------------------------------------------------------- A9
def bce_focal(pt, target, alpha=FOCAL_ALPHA, gamma=FOCAL_GAMMA, eps=FOCAL_EPS):
    """utils/losses.py:83-103 (reduction 'elementwise_mean')."""
    pt = pt.clamp(min=eps, max=1 - eps)
    l = -alpha * (1 - pt) ** gamma * target * torch.log(pt) \
        - (1 - alpha) * pt ** gamma * (1 - target) * torch.log(1 - pt)
    return l.mean()


def permutation_loss(pred, gt):
    """utils/losses.py:419-455."""
    pred = pred.to(torch.float32)
    if pred.dim() == 3:
        pred = pred.squeeze()
    assert bool(torch.all((pred >= 0) * (pred <= 1)))
    assert bool(torch.all((gt >= 0) * (gt <= 1)))
    return torch.tensor(0.0) + bce_focal(pred, gt)


# --------------------------------------------------------------------------- A5 call sites
def sinkhorn_pair(w):
    """multi_graph_matching.py:467-468,518-522 — rows<=cols ensured by the caller."""
    return _sinkhorn(w, dummy_row=True, max_iter=PAIR_SK_ITER, tau=PAIR_SK_TAU, batched_operation=False)


def _project_sinkhorn(V, ms, n_univ, tau, sk_iter):
    """multi_graph_matching.py:329-353."""
    G = len(ms)
    if all(m == ms[0] for m in ms):
        if ms[0] <= n_univ:
            return _sinkhorn(V.reshape(G, -1, n_univ), dummy_row=True, max_iter=sk_iter, tau=tau,
                             batched_operation=True).reshape(-1, n_univ)
        return _sinkhorn(V.reshape(G, -1, n_univ).transpose(1, 2), dummy_row=True, max_iter=sk_iter, tau=tau,
                         batched_operation=True).transpose(1, 2).reshape(-1, n_univ)
    blocks, start = [], 0
    for m in ms:
        blocks.append(V[start:start + m, :n_univ])
        start += m
    n1 = torch.tensor(ms)
    Up = _sinkhorn(torch.stack(pad_tensor(blocks), dim=0), n1=n1, dummy_row=True, max_iter=sk_iter, tau=tau,
                   batched_operation=True)
    return torch.cat([Up[g, :m, :] for g, m in enumerate(ms)], dim=0)


# --------------------------------------------------------------------------- A6
def gagm(A, W, U0, ms, n_univ=UNIV_SIZE, quad_weight=QUAD_WEIGHT, init_tau=GA_TAU0, min_tau=GA_MIN_TAU,
         max_iter=GA_MGM_ITER, sk_iter=GA_SK_ITER, sk_gamma=GA_GAMMA, tol=GA_TOL, trace=None, max_stages=0, perturb=None):
    """multi_graph_matching.py:300-389 with num_clusters==1 (cluster_M == 1, hung_iter True),
    entered from GA_GM.forward :223-244 (W detached :225).

    ``trace`` (optional dict) receives 'V0' (first-iteration V), 'iters' (per-stage
    iteration counts), 'stages' (projector/tau per stage) and 'states' (U at the end of every stage) for parity tests.  ``max_stages`` > 0 (test hook, not
    in the reference) returns the state after that many stages of the schedule.  ``perturb`` (test hook, not in the
    reference): callable ``U = perturb(U, stage, i)`` applied to every Sinkhorn-stage projection - rounding-sized noise
    injected where a second implementation's projector would round differently (tests/golden/make_golden.py uses it to
    refuse goldens that sit on a rounding edge)."""
    ms = [int(m) for m in ms]
    G = len(ms)
    W = W.detach()
    U = U0
    lastU = torch.zeros_like(U)
    tau = init_tau
    projector = "sinkhorn"
    nstages = 0
    if trace is not None:
        trace.update(iters=[], stages=[], states=[])
    while True:
        for i in range(max_iter):
            lastU2, lastU = lastU, U
            UUt = U @ U.t()
            V = torch.linalg.multi_dot([A, UUt, A, U]) * quad_weight * 2 + W @ U
            V = V / G
            if trace is not None and "V0" not in trace:
                trace["V0"] = V.clone()
            if projector == "hungarian":
                parts, start = [], 0
                for m in ms:
                    parts.append(hungarian(V[start:start + m, :n_univ]))
                    start += m
                U = torch.cat(parts, dim=0)
            else:
                U = _project_sinkhorn(V, ms, n_univ, tau, sk_iter)
                if perturb is not None:
                    U = perturb(U, nstages, i)
            if G == 2:
                U[:ms[0], :] = torch.eye(ms[0], n_univ, dtype=U.dtype)
            if torch.norm(U - lastU) < tol or torch.norm(U - lastU2) == 0:
                break
        if trace is not None:
            trace["iters"].append(i + 1)
            trace["stages"].append((projector, tau))
            trace["states"].append(U.clone())                 # the state the NEXT stage starts from
        nstages += 1
        if projector == "hungarian" or (max_stages and nstages >= max_stages):
            break
        elif tau > min_tau:
            tau *= sk_gamma
        else:
            projector = "hungarian"
    return U


# --------------------------------------------------------------------------- A8
def mgm3_unsup_forward(p, nodes, labels, U, n_univ=UNIV_SIZE, dropout_masks=None, trace=None, forced_U=None):
    """multi_graph_matching.py:487-569 (+ collect_intra_class_matching_wrapper :594-633).
    Returns the scalar loss, or None when fewer than two graphs (:489-490).  ``forced_U`` (test hook, not in the
    reference) replaces the solver's output: teacher-forced comparisons of two implementations on the same pseudo-labels."""
    if nodes is None or len(nodes) == 1:
        return None
    ms = [len(l) for l in labels]
    G = len(ms)
    off = [0] + list(np.cumsum(ms))
    M = off[-1]

    A = torch.zeros(M, M, dtype=nodes[0].dtype)          # float32 on the path; float64 when a test asks for the truth
    for g, x in enumerate(nodes):
        adj = mha_adjacency(p, x, dropout_mask=None if dropout_masks is None else dropout_masks[g])
        A[off[g]:off[g + 1], off[g]:off[g + 1]].add_(adj[:ms[g], :ms[g]])
    A.fill_diagonal_(0)

    Wds = torch.zeros(M, M, dtype=nodes[0].dtype)
    for a in range(G):          # src
        for b in range(G):      # tgt
            if a < b:
                continue
            Wab = affinity(p, nodes[a], nodes[b])[:ms[a], :ms[b]]
            if ms[b] >= ms[a]:
                Wab_ds = sinkhorn_pair(Wab)
            else:
                Wab_ds = sinkhorn_pair(Wab.t()).t()
            Wds[off[a]:off[a + 1], off[b]:off[b + 1]] += Wab_ds
            if a != b:
                Wds[off[b]:off[b + 1], off[a]:off[a + 1]] += Wab_ds.t()

    U0 = torch.cat([x @ U.t() for x in nodes], dim=0).detach()
    Ub = gagm(A, Wds, U0, ms, n_univ, trace=trace) if forced_U is None else forced_U.to(Wds.dtype)
    Ul = [Ub[off[g]:off[g + 1]] for g in range(G)]
    if trace is not None:
        trace.update(A=A.detach().clone(), Wds=Wds.detach().clone(), U0=U0.clone(), Ub=Ub.clone())

    loss, npairs = 0, 0
    for i, j in itertools.combinations(range(G), 2):
        if ms[j] >= ms[i]:
            s = Wds[off[i]:off[i + 1], off[j]:off[j + 1]]
        else:
            s = Wds[off[j]:off[j + 1], off[i]:off[i + 1]].t()
        x_gt = Ul[i] @ Ul[j].t()
        loss = loss + permutation_loss(s.unsqueeze(0), x_gt.unsqueeze(0))
        npairs += 1
    return loss / npairs


# --------------------------------------------------------------------------- N3 (source-training matching loss)
HIPPI_MAX_ITER, HIPPI_SK_ITER, HIPPI_SK_TAU, HIPPI_TOL = 50, 20, 1 / 200., 1e-5
USUP_LOSS_W, USUP_LOSS_LAM = 0.1, 1e-4


def mha_full(p, x, prefix="Net_U.g_gene.", attn_mask=None, out_mask=None):
    """utils/attentions.py:60-86 (version 'v2', 1 head) returning BOTH results: output = LayerNorm(x + dropout(
    linear_final(attention @ v))) and the attention map.  ``attn_mask`` (n,n) / ``out_mask`` (n,256) of 0/1 reproduce
    train-mode dropout draws (attentions.py:40,85); None = eval mode."""
    d = x.shape[1]
    k = F.linear(x, p[prefix + "linear_k.weight"], p[prefix + "linear_k.bias"])
    v = F.linear(x, p[prefix + "linear_v.weight"], p[prefix + "linear_v.bias"])
    q = F.linear(x, p[prefix + "linear_q.weight"], p[prefix + "linear_q.bias"])
    att = torch.softmax((q @ k.t()) * (d ** -0.5), dim=1)
    if attn_mask is not None:
        att = att * attn_mask / 0.9
    out = F.linear(att @ v, p[prefix + "linear_final.weight"], p[prefix + "linear_final.bias"])
    if out_mask is not None:
        out = out * out_mask / 0.9
    out = F.layer_norm(x + out, (d,), p[prefix + "layer_norm.weight"], p[prefix + "layer_norm.bias"])
    return out, att


def g_universe(p, nodes, U):
    """G_Universe.forward, multi_graph_matching.py:90-112 (+ cos_similarity :114-117).  N = cat(node_g U^T); the edge
    list is attention / (D + 1e-8) with D = 1 - sum(x^2)/||x||^2, which is zero up to fp32 rounding: the edges are
    ~1e7-scaled with rounding-decided signs."""
    N_list, E_list = [], []
    for x in nodes:
        node, edge = mha_full(p, x)
        norms = torch.norm(node, p=2, dim=1, keepdim=True)
        D = 1 - torch.sum(node * node, dim=1, keepdim=True) / norms ** 2
        E_list.append(edge * (1 / (D + 1e-8)))
        N_list.append(node @ U.t())
    return torch.cat(N_list, dim=0), E_list


def hippi(W, U0, ms, d=UNIV_SIZE, projector="sinkhorn", max_iter=HIPPI_MAX_ITER, sk_iter=HIPPI_SK_ITER, tau=HIPPI_SK_TAU,
          trace=None):
    """HiPPI.forward, multi_graph_matching.py:414-449: V = (W U) U^T (W U) (chain_matmul = cheapest association),
    per-graph Sinkhorn (dummy rows, tau 1/200, 20 sweeps) or Hungarian, stop when ||U - lastU||_F < 1e-5."""
    ms = [int(m) for m in ms]
    U = U0
    for i in range(max_iter):
        lastU = U
        WU = W @ U
        V = torch.linalg.multi_dot([WU, U.t(), WU])
        if trace is not None and "V0" not in trace:
            trace["V0"] = V.clone()
        parts, start = [], 0
        for m in ms:
            blk = V[start:start + m, :d]
            if projector == "sinkhorn":
                parts.append(_sinkhorn(blk, dummy_row=True, max_iter=sk_iter, tau=tau, batched_operation=False))
            elif projector == "hungarian":
                parts.append(hungarian(blk))
            else:
                raise NameError("Unknown projector {}.".format(projector))
            start += m
        U = torch.cat(parts, dim=0)
        if torch.norm(U - lastU) < HIPPI_TOL:
            break
    if trace is not None:
        trace["iters"] = i + 1
    return U


def label_block_matrix(labels, num_classes):
    """U_sup.forward :146-152 with build_label_wise/one_hot :161-166: W[a,b] = 1 iff the two nodes carry the same label."""
    oh = torch.cat([torch.eye(num_classes)[l.long() - 1, :] for l in labels], dim=0)
    return oh @ oh.t()


def u_sup_forward(p, nodes, labels, num_classes=2, n_univ=UNIV_SIZE, forced_target=None, trace=None):
    """U_sup.forward + U_loss, multi_graph_matching.py:136-169.  ``p`` = U_sup state dict (``U`` + ``Net_U.*``).
    ``forced_target`` replaces the HiPPI result (which is detached in the loss, :156-158), the way the parity tests pin
    everything that carries gradient independently of the rounding-driven edge weights."""
    ms = [len(l) for l in labels]
    N, edges = g_universe(p, nodes, p["U"])
    Us = _sinkhorn(N, max_iter=PAIR_SK_ITER, tau=PAIR_SK_TAU, batched_operation=False)
    if forced_target is None:
        A = torch.block_diag(*edges)
        Wl = label_block_matrix(labels, num_classes)
        A_ = Wl.t() @ A @ Wl
        target = hippi(A_, Us, ms, n_univ)
    else:
        target = forced_target
    if trace is not None:
        trace.update(N=N.detach().clone(), Us=Us.detach().clone(), target=target.detach().clone())
    return USUP_LOSS_W * F.mse_loss(Us, target.detach()) + USUP_LOSS_LAM * torch.norm(p["U"], p="fro")


# --------------------------------------------------------------------------- A2
STRIDES = (4, 8, 16, 32, 64)
SIZE_RANGES = ((-1, 64), (64, 128), (128, 256), (256, 512), (512, 100000000))
INF = 100000000


def level_locations(h, w, stride):
    """build_graph.py:144-157 — (x, y) = (j*s + s//2, i*s + s//2), raster order."""
    ys, xs = torch.meshgrid(torch.arange(0, h * stride, stride, dtype=torch.float32),
                            torch.arange(0, w * stride, stride, dtype=torch.float32), indexing="ij")
    return torch.stack((xs.reshape(-1), ys.reshape(-1)), dim=1) + stride // 2


def location_labels(locs, ranges, boxes, classes):
    """build_graph.py:70-115 for one image with >=1 box: label per location (0 = none)."""
    xs, ys = locs[:, 0], locs[:, 1]
    area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    ltrb = torch.stack([xs[:, None] - boxes[:, 0][None], ys[:, None] - boxes[:, 1][None],
                        boxes[:, 2][None] - xs[:, None], boxes[:, 3][None] - ys[:, None]], dim=2)
    inside = ltrb.min(dim=2)[0] > 0
    mx = ltrb.max(dim=2)[0]
    cared = (mx >= ranges[:, [0]]) & (mx <= ranges[:, [1]])
    a = area[None].repeat(len(locs), 1)
    a[inside == 0] = INF
    a[cared == 0] = INF
    amin, ind = a.min(dim=1)
    lab = (classes + 1)[ind]
    lab[amin == INF] = 0
    return lab


def prototype_computation(features, boxes_per_img, classes_per_img, sample_dist=SAMPLE_DIST):
    """build_graph.py:160-250.  ``features``: 5 NCHW maps; ``boxes_per_img``/``classes_per_img``:
    per image (k,4) xyxy float / (k,) int64 (k may be 0).  Reproduces the reference's
    image-index quirk: images without boxes are skipped when labels are built (:79) but
    features are indexed by list position (:173-181)."""
    if not any(len(b) for b in boxes_per_img):
        return None, None
    locs = [level_locations(f.shape[-2], f.shape[-1], STRIDES[l]) for l, f in enumerate(features)]
    npts = [len(x) for x in locs]
    ranges = torch.cat([torch.tensor(SIZE_RANGES[l], dtype=torch.float32)[None].expand(n, -1)
                        for l, n in enumerate(npts)], dim=0)
    allp = torch.cat(locs, dim=0)
    labels = []
    for bx, cl in zip(boxes_per_img, classes_per_img):
        if len(bx):
            labels.append(torch.split(location_labels(allp, ranges, bx, cl), npts, dim=0))
    C = features[0].shape[1]
    nodes_b, labels_b = [], []
    for b in range(len(labels)):
        pts, lbs = [], []
        for l, lab in enumerate(labels[b]):
            feat = features[l][b].permute(1, 2, 0).reshape(-1, C)
            pos = lab.reshape(-1) > 0
            fa, la = feat[pos], lab.reshape(-1)[pos]
            step = len(la) // sample_dist
            if step > 1:
                fa, la = fa[::step], la[::step]
            pts.append(fa)
            lbs.append(la)
        nodes_b.append(torch.cat(pts, dim=0))
        labels_b.append(torch.cat(lbs, dim=0))
    return nodes_b, labels_b


# --------------------------------------------------------------------------- A11
def sgd_step(params, grads, bufs, lr, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD as configured by detectron2 build_optimizer [3P]
    (train_net.py:65, trainer.py:480-482): d = g + wd*p; buf = d on first use else
    mom*buf + d; p -= lr*buf.  Tensors whose grad is None are skipped.  In place."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        d = g + weight_decay * p if weight_decay != 0 else g
        if bufs[i] is None:
            bufs[i] = d.clone()
        else:
            bufs[i].mul_(momentum).add_(d)
        p.sub_(lr * bufs[i])


# --------------------------------------------------------------------------- A12
def dice_scores(pred_masks, pred_classes, pred_scores, gt_masks, gt_classes, thres=0.9):
    """evaluation/dice_metric.py:25-78 (Dice part): per kept prediction, best Dice over
    same-class GT, x100."""
    keep = pred_scores >= thres
    out = []
    for pc, pm in zip(pred_classes[keep], pred_masks[keep]):
        best = 0
        for gc, gm in zip(gt_classes, gt_masks):
            if pc == gc:
                inter = np.logical_and(pm, gm).sum()
                best = max(best, 2 * inter / (pm.sum() + gm.sum() + 1e-6))
        out.append(best * 100)
    return out
