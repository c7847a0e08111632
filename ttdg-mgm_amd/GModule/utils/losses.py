"""BCEFocalLoss / PermutationLoss — mirror of reference utils/losses.py:72-103, 400-455.

Stand-alone modules for API compatibility (element-wise torch ops on the caller's device); on the TTA path
the loss and its gradient come from the fused kernel csrc/loss.hip via ops.MatchingLossFn."""
import torch
import torch.nn as nn


class BCEFocalLoss(nn.Module):
    def __init__(self, gamma=2, alpha=0.25, reduction='elementwise_mean'):
        super().__init__()
        self.gamma, self.alpha, self.reduction = gamma, alpha, reduction

    def forward(self, _input, target):
        eps = 1e-6
        p = _input.clamp(min=eps, max=1 - eps)
        pos = -self.alpha * (1 - p) ** self.gamma * target * torch.log(p)
        neg = -(1 - self.alpha) * p ** self.gamma * (1 - target) * torch.log(1 - p)
        l = pos + neg
        if self.reduction == 'elementwise_mean':
            return l.mean()
        if self.reduction == 'sum':
            return l.sum()
        raise ValueError("unsupported reduction {}".format(self.reduction))


class PermutationLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.loss = BCEFocalLoss()

    def forward(self, pred_dsmat, gt_perm, src_ns=None, tgt_ns=None):
        pred = pred_dsmat.to(dtype=torch.float32)
        if pred.dim() == 3:
            pred = pred.squeeze()
        assert torch.all((pred >= 0) * (pred <= 1))
        assert torch.all((gt_perm >= 0) * (gt_perm <= 1))
        return torch.tensor(0.).to(pred.device) + self.loss(pred, gt_perm)
