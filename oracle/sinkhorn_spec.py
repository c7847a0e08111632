"""Log-space Sinkhorn — oracle restatement (TEST INFRASTRUCTURE, see oracle/__init__.py).

Stands in for ``pygmtools.sinkhorn(..., backend='pytorch')`` (pygmtools==0.3.8,
reference requirements.txt:66), which the reference calls from
adapteacher/modeling/GModule/utils/sinkhorn.py:85-87.  pygmtools is not
installed in this image and is not vendored by the reference, so this file
restates the published algorithm as written down in SURVEY.md Appendix B.

**PARITY UNPINNED** with respect to the real package: no reference test or
golden vector pins this boundary.  It is pinned by property
tests only (tests/test_oracle_sinkhorn.py).

Signature mirrors the call the reference makes:
    sinkhorn(s, n1=nrows, n2=ncols, dummy_row=..., max_iter=..., tau=...,
             batched_operation=..., backend='pytorch')
"""
import torch

NEG_INF = float("-inf")
DUMMY_FILL = -100.0  # value written into dummy rows *after* tau scaling (Appendix B step 5)


def _as_long(v, b, default, device):
    if v is None:
        return torch.full((b,), int(default), dtype=torch.long, device=device)
    return torch.as_tensor(v, device=device).to(torch.long).reshape(b).clone()


def sinkhorn(s, n1=None, n2=None, unmatch1=None, unmatch2=None, dummy_row=False,
             max_iter=10, tau=1.0, batched_operation=False, backend="pytorch"):
    assert unmatch1 is None and unmatch2 is None, "unmatch weights are not on the TTA path"
    squeeze = s.dim() == 2
    if squeeze:
        s = s.unsqueeze(0)
    assert s.dim() == 3
    b, r, c = s.shape
    dev = s.device

    # step 1: whole-batch orientation, rows <= cols
    flipped = c < r
    if flipped:
        s = s.transpose(1, 2)
        n1, n2 = n2, n1
        r, c = c, r
    # step 2
    nr = _as_long(n1, b, r, dev)
    nc = _as_long(n2, b, c, dev)

    # step 3: per-matrix orientation inside the (r, c) frame
    per_flip = nr > nc
    if bool(per_flip.any()):
        st = s.transpose(1, 2)[:, :r, :]
        st = torch.cat([st, s.new_full((b, r, c - r), NEG_INF)], dim=2)
        s = torch.where(per_flip.view(b, 1, 1), st, s)
        nr, nc = torch.where(per_flip, nc, nr), torch.where(per_flip, nr, nc)

    # step 4
    log_s = s / tau

    # step 5: dummy rows make every valid block square
    n_extra = 0
    if dummy_row:
        n_extra = c - r
        ori_nr = nr
        nr = nc.clone()
        log_s = torch.cat([log_s, log_s.new_full((b, n_extra, c), NEG_INF)], dim=1)
        for k in range(b):
            log_s[k, int(ori_nr[k]):int(nr[k]), :int(nc[k])] = DUMMY_FILL

    R = log_s.shape[1]
    # step 6: alternating normalisation
    if batched_operation:
        rmask = torch.arange(R, device=dev).view(1, R, 1) < nr.view(b, 1, 1)
        cmask = torch.arange(c, device=dev).view(1, 1, c) < nc.view(b, 1, 1)
        log_s = torch.where(rmask & cmask, log_s, log_s.new_full((), NEG_INF))
        for it in range(max_iter):
            if it % 2 == 0:
                lse = torch.logsumexp(log_s, 2, keepdim=True)
                log_s = log_s - torch.where(rmask, lse, torch.zeros_like(lse))
            else:
                lse = torch.logsumexp(log_s, 1, keepdim=True)
                log_s = log_s - torch.where(cmask, lse, torch.zeros_like(lse))
        out = log_s
    else:
        out = log_s.new_full((b, R, c), NEG_INF)
        for k in range(b):
            blk = log_s[k, :int(nr[k]), :int(nc[k])]
            for it in range(max_iter):
                if it % 2 == 0:
                    blk = blk - torch.logsumexp(blk, 1, keepdim=True)
                else:
                    blk = blk - torch.logsumexp(blk, 0, keepdim=True)
            out[k, :int(nr[k]), :int(nc[k])] = blk

    # step 7: drop dummy rows
    if dummy_row:
        if n_extra > 0:
            out = out[:, :-n_extra]
        for k in range(b):
            out[k, int(ori_nr[k]):int(nr[k]), :int(nc[k])] = NEG_INF

    # step 8: undo the orientations
    if bool(per_flip.any()):
        ot = out.transpose(1, 2)[:, :out.shape[1], :]
        ot = torch.cat([ot, out.new_full((b, out.shape[1], out.shape[2] - out.shape[1]), NEG_INF)], dim=2)
        out = torch.where(per_flip.view(b, 1, 1), ot, out)
    if flipped:
        out = out.transpose(1, 2)
    res = torch.exp(out)
    return res.squeeze(0) if squeeze else res
