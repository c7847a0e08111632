"""Affinity — mirror of reference utils/affinity.py:9-57.

Same parameters (fc_M.0 512x512+512, fc_M.2 1x512+1, project_sr / project_tg 256x256, no bias) and the
same initialisation; the forward is the decomposed form
    M_ij = w2 . relu(P_i + Q_j) + b2,   P = (X Psr^T) W1[:, :d]^T,   Q = (Y Ptg^T) W1[:, d:]^T + b1
on the MFMA GEMM + the pairwise VALU kernel, instead of an (N1, N2, 512) tensor through an MLP."""
import torch
import torch.nn as nn

from ... import ops


class _AffinityFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Y, W1, b1, w2, b2, Psr, Ptg):
        ops.check_f32(X, Y, W1, b1, w2, b2, Psr, Ptg)
        n1, n2, d, H = X.shape[0], Y.shape[0], X.shape[1], W1.shape[0]
        gr = ops.graphs([n2, n1])            # graph 0 = Y (columns), graph 1 = X (rows): block (1,0) is wanted
        Xs, Yt = ops.linear_raw(X, Psr), ops.linear_raw(Y, Ptg)
        P = torch.zeros(n1 + n2, H, device=X.device)
        Q = torch.zeros(n1 + n2, H, device=X.device)
        ops.linear_raw(Xs, W1, None, 0, H, out=P[n2:])
        ops.linear_raw(Yt, W1, b1, d, H, out=Q[:n2])
        w2f = w2.reshape(-1)
        part = ops.affinity_pairwise_fwd(P, Q, w2f, gr, 1)
        ctx.save_for_backward(X, Y, Xs, Yt, P, Q, W1, w2f, Psr, Ptg)
        return part[0, n2:, :n2] + b2

    @staticmethod
    def backward(ctx, dMblk):
        X, Y, Xs, Yt, P, Q, W1, w2f, Psr, Ptg = ctx.saved_tensors
        n1, n2, d, H = X.shape[0], Y.shape[0], X.shape[1], W1.shape[0]
        gr = ops.graphs([n2, n1])
        dM = torch.zeros(n1 + n2, n1 + n2, device=X.device)
        dM[n2:, :n2] = dMblk
        dP, dQ, dw2, db2 = ops.affinity_pairwise_bwd(P, Q, w2f, dM, gr)
        dPx, dQy = dP[n2:].contiguous(), dQ[:n2].contiguous()
        dW1 = torch.empty_like(W1)
        ops.gemm(dPx, 1, H, Xs, 1, d, dW1, H, 1, H, d, n1)
        ops.gemm(dQy, 1, H, Yt, 1, d, dW1, H, 1, H, d, n2, c_off=d)
        db1 = ops.colsum(dQy)
        dXs, dYt = torch.empty_like(Xs), torch.empty_like(Yt)
        ops.gemm(dPx, H, 1, W1, 1, H, dXs, d, 1, n1, d, H)
        ops.gemm(dQy, H, 1, W1, 1, H, dYt, d, 1, n2, d, H, b_off=d)
        dPsr, dPtg = torch.empty_like(Psr), torch.empty_like(Ptg)
        ops.gemm(dXs, 1, d, X, 1, d, dPsr, d, 1, d, d, n1)
        ops.gemm(dYt, 1, d, Y, 1, d, dPtg, d, 1, d, d, n2)
        dX, dY = torch.empty_like(X), torch.empty_like(Y)
        ops.gemm(dXs, d, 1, Psr, 1, d, dX, d, 1, n1, d, d)
        ops.gemm(dYt, d, 1, Ptg, 1, d, dY, d, 1, n2, d, d)
        return dX, dY, dW1, db1, dw2.view(1, H), db2, dPsr, dPtg


class Affinity(nn.Module):
    def __init__(self, d=256):
        super().__init__()
        self.d = d
        self.fc_M = nn.Sequential(nn.Linear(2 * d, 2 * d), nn.ReLU(), nn.Linear(2 * d, 1))
        self.project_sr = nn.Linear(d, d, bias=False)
        self.project_tg = nn.Linear(d, d, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        for layer in (self.fc_M[0], self.fc_M[2]):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)
        nn.init.normal_(self.project_sr.weight, std=0.01)
        nn.init.normal_(self.project_tg.weight, std=0.01)

    def forward(self, X, Y):
        M = _AffinityFn.apply(X.contiguous(), Y.contiguous(), self.fc_M[0].weight, self.fc_M[0].bias,
                              self.fc_M[2].weight, self.fc_M[2].bias, self.project_sr.weight, self.project_tg.weight)
        return M.squeeze()
