"""pad_tensor — mirror of reference utils/pad_tensor.py:5-31 (zero-pad a list of tensors to a common shape).
Pure shape plumbing (torch pad on whatever device the inputs live on)."""
import torch.nn.functional as F


def pad_tensor(inp):
    assert len(inp) > 0 and all(hasattr(t, "shape") for t in inp)
    nd = inp[0].dim()
    target = [max(int(t.shape[d]) for t in inp) for d in range(nd)]
    out = []
    for t in inp:
        pad = []
        for d in reversed(range(nd)):
            pad.extend((0, target[d] - int(t.shape[d])))
        out.append(F.pad(t, tuple(pad), "constant", 0))
    return out
