"""hungarian — mirror of reference utils/hungarian.py:8-66.

The reference copies every matrix to the host and calls scipy.optimize.linear_sum_assignment;
here the LAP is solved on the GPU, one wavefront per matrix, with scipy's exact tie rules
(csrc/lap_device.h)."""
import torch

from ... import ops


def hungarian(s, n1=None, n2=None, nproc=1):
    """s: (n1, n2) or (b, n1, n2) scores to MAXIMISE -> 0/1 matrix of the same shape.
    ``n1``/``n2`` (per-matrix valid sizes) restrict the solve to the leading block, as the reference does;
    ``nproc`` is accepted for signature compatibility (the batch is already parallel on the device)."""
    if s.dim() == 2:
        s3, squeeze = s.unsqueeze(0), True
    elif s.dim() == 3:
        s3, squeeze = s, False
    else:
        raise ValueError("input data shape not understood: {}".format(s.shape))
    s3 = s3.detach()
    if n1 is None and n2 is None:
        x = ops.lap_batched(s3)
    else:
        b = s3.shape[0]
        r = [int(v) for v in n1] if n1 is not None else [s3.shape[1]] * b
        c = [int(v) for v in n2] if n2 is not None else [s3.shape[2]] * b
        x = torch.zeros_like(s3, dtype=torch.float32)
        for i in range(b):   # ragged batch: one launch per distinct valid block
            x[i, :r[i], :c[i]] = ops.lap_batched(s3[i:i + 1, :r[i], :c[i]])[0]
    return x.squeeze(0) if squeeze else x
