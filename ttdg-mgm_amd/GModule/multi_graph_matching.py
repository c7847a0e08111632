"""GA_GM / U_sup / MGM3_unsup — mirror of reference multi_graph_matching.py for the TTA path.

Class names, constructor signatures, hyper-parameters and state-dict keys follow the reference
(:119-134, :191-221, :451-474); the forward of MGM3_unsup is one fused device pipeline
(ops.MatchingLossFn) instead of per-pair Python loops, and GA_GM.forward is one persistent kernel
instead of up to 1200 iterations of small launches and host round-trips."""
import os

import torch
import torch.nn as nn

from .. import ops
from .utils.affinity import Affinity
from .utils.attentions import MultiHeadAttention
from .utils.losses import PermutationLoss
from .utils.sinkhorn import Sinkhorn


class GA_GM(nn.Module):
    """Graduated-assignment multi-graph matching (reference :191-389), num_clusters == 1 only
    (the only mode MGM3_unsup uses, :533 -> :243-244)."""

    def __init__(self, mgm_iter=(200,), cluster_iter=10, sk_iter=20, sk_tau0=(0.5,), sk_gamma=0.5,
                 cluster_beta=(1., 0.), converge_tol=1e-5, min_tau=(1e-2,), projector0=('sinkhorn',)):
        super().__init__()
        self.mgm_iter, self.cluster_iter, self.sk_iter = mgm_iter, cluster_iter, sk_iter
        self.sk_tau0, self.sk_gamma, self.cluster_beta = sk_tau0, sk_gamma, cluster_beta
        self.converge_tol, self.min_tau, self.projector0 = converge_tol, min_tau, projector0
        self.last_info = None

    def _cfg(self, quad_weight):
        if self.projector0[0] != 'sinkhorn':
            raise NameError('Unknown projecter name: {}'.format(self.projector0[0]))
        return ops.gagm_cfg(tau0=self.sk_tau0[0], gamma=self.sk_gamma, min_tau=self.min_tau[0], tol=self.converge_tol,
                            quad_weight=quad_weight, max_iter=self.mgm_iter[0], sk_iter=self.sk_iter,
                            profile=bool(os.environ.get("TTDG_GAGM_PROFILE")))

    def solve_packed(self, apack, W, U0, sizes, quad_weight=1.):
        U, info, V0 = ops.gagm_solve(apack, W.detach().contiguous(), U0.detach().contiguous(), ops.graphs(sizes), sizes,
                                     self._cfg(quad_weight))
        self.last_info = info
        return U

    def forward(self, A, W, U0, ms, n_univ, quad_weight=1., cluster_quad_weight=1., num_clusters=1):
        """A: (M,M) block-diagonal adjacency, W: (M,M), U0: (M,n_univ), ms: (G,) node counts.
        Returns (U, cluster) with cluster = zeros(G) as the reference does for num_clusters == 1."""
        if num_clusters != 1:
            raise NotImplementedError("multi-cluster matching is not on the TTA path (reference :243-244)")
        if n_univ != ops.UNIV:
            raise ValueError("universe size is fixed at {} on this path".format(ops.UNIV))
        sizes = [int(m) for m in ms]
        off, blocks = 0, []
        for n in sizes:
            blocks.append(A[off:off + n, off:off + n].detach().reshape(-1))
            off += n
        U = self.solve_packed(torch.cat(blocks).contiguous().float(), W.float(), U0.float(), sizes, quad_weight)
        return U, torch.zeros(len(sizes), dtype=torch.int)


class _FeatGraphParams(nn.Module):
    """Parameter holder with the names of utils/graph_network.py:95-99 (Feat2Graph)."""

    def __init__(self, d):
        super().__init__()
        self.wq = nn.Linear(d, d)
        self.wk = nn.Linear(d, d)


class _UniverseNetParams(nn.Module):
    """Parameter holder with the names of G_Universe (reference :77-88); training-time only."""

    def __init__(self, dim, univ_size):
        super().__init__()
        self.f2g = _FeatGraphParams(dim)
        self.g_gene = MultiHeadAttention(dim, 1, dropout=0.1, version='v2')
        self.adapt = nn.Linear(dim, dim)
        self.affinity_layer = Affinity(dim)
        self.univ_size = univ_size


class U_sup(nn.Module):
    """Holder of the learned universe ``U`` (reference :119-134).  The TTA path only READS ``.U``
    (rcnn.py:353); the supervised HiPPI loss of ``forward`` belongs to source training.
    All reference parameters are declared so that checkpoints load with strict=True."""

    def __init__(self, num_cls, univ_size, dim=256):
        super().__init__()
        self.univ_size = univ_size
        self.U = nn.Parameter(torch.randn(univ_size, dim) + 1 / self.univ_size)
        self.num_classes = num_cls
        self.Net_U = _UniverseNetParams(dim, univ_size)
        self.node_affinity = Affinity(256)
        self.sinkhorn = Sinkhorn(max_iter=20, tau=0.05, epsilon=1e-10, batched_operation=False)

    def forward(self, nodes, labels):
        raise NotImplementedError("U_sup.forward is the source-training matching loss (SURVEY.md §8f N3); "
                                  "test-time adaptation only reads U_sup.U")


class MGM3_unsup(nn.Module):
    """Unsupervised multi-graph-matching loss used for test-time adaptation (reference :451-633)."""

    def __init__(self, num_cls, univ_size, dim=256):
        super().__init__()
        self.univ_size = univ_size
        self.num_classes = num_cls
        self.quad_weight = 0.5
        self.cluster_quad_weight = 1
        self.perm_loss = 'perm'
        self.node_affinity = Affinity(d=dim)
        self.intra_domain_graph = MultiHeadAttention(dim, 1, dropout=0.1, version='v2')
        self.sinkhorn = Sinkhorn(max_iter=20, tau=0.05, epsilon=1e-10, batched_operation=False)
        self.ga_mgmc = GA_GM(mgm_iter=[200], cluster_iter=10, sk_iter=20, sk_tau0=[0.1], sk_gamma=0.5,
                             cluster_beta=[1.0, 0.0], converge_tol=1.0e-3, min_tau=[1.0e-2],
                             projector0=['sinkhorn', 'sinkhorn'])
        self.criterion = PermutationLoss()
        self.dropout_seed = 0          # Philox key of the train-mode attention dropout; bumped every call
        self.check_range = False       # True: sync and raise like losses.py:437-442 when Wds leaves [0,1]
        self.keep_trace = False        # True: keep the intermediates of the last forward in ``self.last``
        self.last = None

    def forward(self, nodes, labels, U, trace=None, forced_U=None):
        """nodes: list of (n_g, dim) tensors, labels: list of (n_g,) -> scalar loss, or None when there are
        fewer than two graphs (reference :489-490).  ``trace`` (dict) receives the intermediates;
        ``forced_U`` replaces the solver's output (teacher-forced parity tests)."""
        if nodes is None or len(nodes) == 1:
            return None
        sizes = [len(l) for l in labels]
        if sizes != [int(x.shape[0]) for x in nodes]:
            raise ValueError("nodes and labels disagree on the graph sizes")
        if 0 in sizes:
            # a detection too small to contain any FPN point yields an empty graph; the reference's loops are
            # undefined there (0-row Sinkhorn / LAP).  Documented deviation: such graphs take no part in the matching.
            keep = [g for g, n in enumerate(sizes) if n > 0]
            nodes, labels, sizes = [nodes[g] for g in keep], [labels[g] for g in keep], [sizes[g] for g in keep]
            if len(sizes) < 2:
                return None
        if trace is None and self.keep_trace:
            trace = {}
        X = torch.cat(list(nodes), dim=0).float().contiguous()
        aff, att = self.node_affinity, self.intra_domain_graph
        self.dropout_seed += 1
        opts = {
            "pair_tau": self.sinkhorn.tau, "pair_iters": self.sinkhorn.max_iter,
            "drop_p": att.drop_p if self.training else 0.0, "seed": self.dropout_seed,
            "gagm_cfg": self.ga_mgmc._cfg(self.quad_weight), "trace": trace, "forced_U": forced_U,
        }
        loss, flag = ops.MatchingLossFn.apply(
            X, aff.fc_M[0].weight, aff.fc_M[0].bias, aff.fc_M[2].weight, aff.fc_M[2].bias,
            aff.project_sr.weight, aff.project_tg.weight,
            att.linear_q.weight, att.linear_q.bias, att.linear_k.weight, att.linear_k.bias,
            U.detach().contiguous(), sizes, opts)
        if trace is not None and trace.get("info") is not None:
            self.ga_mgmc.last_info = trace["info"]
        if self.keep_trace:
            trace["X"], trace["sizes"] = X.detach(), sizes
            self.last = trace
        if self.check_range:
            assert int(flag.item()) == 0, "pred_dsmat / gt_perm left [0, 1]"
        return loss

    def one_hot(self, x):
        return torch.eye(self.num_classes)[x.long().cpu() - 1, :].to(x.device)
