"""Sinkhorn — mirror of reference utils/sinkhorn.py:7-87 (log-space forward).

The reference delegates to pygmtools.sinkhorn (un-vendored third party); the arithmetic implemented by
csrc/sinkhorn.hip follows SURVEY.md Appendix B.  The differentiable use on the TTA path (pair stage) is
fused into ops.MatchingLossFn; this stand-alone module is the general operator: gradient-free as GA_GM uses it, or
differentiable (ops.SinkhornFn: the backward rebuilds every sweep from the logged dual potentials) when its input
requires grad - up to 64 sweeps."""
import torch
import torch.nn as nn

from ... import ops


class Sinkhorn(nn.Module):
    def __init__(self, max_iter=10, tau=1., epsilon=1e-4, log_forward=True, batched_operation=False):
        super().__init__()
        self.max_iter = max_iter
        self.tau = tau
        self.epsilon = epsilon
        self.log_forward = log_forward
        if not log_forward:
            print('Warning: Sinkhorn algorithm without log forward is deprecated because log_forward is more stable.')
        self.batched_operation = batched_operation   # same arithmetic either way on the device

    def forward(self, s, nrows=None, ncols=None, dummy_row=False):
        if not self.log_forward:
            raise NotImplementedError("only the log-space forward (the one on the TTA path) is implemented")
        squeeze = s.dim() == 2
        s3 = s.unsqueeze(0) if squeeze else s
        if s.requires_grad and torch.is_grad_enabled():
            if self.max_iter > 64:
                raise NotImplementedError("differentiable Sinkhorn logs the potentials of at most 64 sweeps")
            out = ops.SinkhornFn.apply(s3.float(), nrows, ncols, bool(dummy_row), float(self.tau), int(self.max_iter))
        else:
            out = ops.sinkhorn_batched(s3.float(), nrows, ncols, dummy_row, self.tau, self.max_iter)
        return out.squeeze(0) if squeeze else out
