"""FusedSGD — the optimizer step of the TTA loop (reference engine/trainer.py:480-482) as ONE kernel launch.

Semantics are torch.optim.SGD's as detectron2's build_optimizer configures it [3P] (momentum 0.9, per-group
weight decay, no dampening / nesterov): parameters whose ``.grad`` is None are skipped, the momentum buffer is
initialised with the first decayed gradient.  The per-step descriptor table (pointers change as autograd
allocates fresh gradients) is staged in pinned host memory and copied asynchronously."""
import ctypes as C

import torch

from . import _lib
from ._lib import call, ptr, stream

CHUNK = 65536  # elements per workgroup


def _same_storage_order(p, t):
    """True when ``t`` walks memory in the same element order as the dense parameter ``p``."""
    if t.stride() == p.stride():
        return True
    if p.is_contiguous() and t.is_contiguous():
        return True
    return p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and t.is_contiguous(memory_format=torch.channels_last)


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0):
        if momentum <= 0:
            raise ValueError("FusedSGD implements SGD with momentum (the TTA configuration)")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._pinned = [None, None]       # two staging buffers: a buffer is rewritten only after its H2D copy completed
        self._copied = [None, None]
        self._flip = 0

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used on the TTA path")
        lr = mom = None
        todo = []
        for group in self.param_groups:
            lr = group["lr"] if lr is None else lr
            mom = group["momentum"] if mom is None else mom
            if group["lr"] != lr or group["momentum"] != mom:
                raise ValueError("FusedSGD needs one lr / momentum for all groups (true for the TTA optimizer)")
            for p in group["params"]:
                if p.grad is None:
                    continue
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if p.dtype != torch.float32 or not dense:
                    raise TypeError("FusedSGD handles dense float32 parameters")
                # the update is element-wise: any dense layout works as long as p, grad and buffer share it
                # (same storage order, not same stride tuple: a 1 x 1 filter is contiguous AND channels-last with two different tuples)
                g = p.grad if _same_storage_order(p, p.grad) else torch.empty_like(p).copy_(p.grad)
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first:
                    st["momentum_buffer"] = torch.empty_like(p)       # preserves p's strides
                elif not _same_storage_order(p, st["momentum_buffer"]):
                    # the parameter changed layout after the buffer was made (the backbone moves its filters to channels-last on its
                    # first fp32 GPU forward; ``load_state_dict`` brings buffers in the layout they were saved in): re-lay the buffer,
                    # or the kernel would pair p[i] with the momentum of another element
                    st["momentum_buffer"] = torch.empty_like(p).copy_(st["momentum_buffer"])
                    self._chunk_key = None
                todo.append((p, g, st["momentum_buffer"], group["weight_decay"], first))
        if not todo:
            return None
        dev = todo[0][0].device
        nt = len(todo)
        # the chunk map only depends on WHICH tensors carry a gradient (the same ~65 on every TTA step): built once per set
        key = tuple(id(t[0]) for t in todo)
        if getattr(self, "_chunk_key", None) != key:
            chunks_t, chunks_o = [], []
            for ti, (p, _, _, _, _) in enumerate(todo):
                n = p.numel()
                for off in range(0, n, CHUNK):
                    chunks_t.append(ti)
                    chunks_o.append(off)
            self._chunk_key, self._chunks = key, (chunks_t, chunks_o)
        chunks_t, chunks_o = self._chunks
        nc = len(chunks_t)
        tbytes = C.sizeof(_lib.SgdTensor) * nt
        total = tbytes + 4 * nc + 8 * nc + 64
        slot = self._flip
        self._flip ^= 1
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()
        if self._pinned[slot] is None or self._pinned[slot].numel() < total:
            self._pinned[slot] = torch.empty(total * 2, dtype=torch.uint8).pin_memory()
        host = self._pinned[slot]
        table = (_lib.SgdTensor * nt).from_address(host.data_ptr())
        for ti, (p, g, b, wd, first) in enumerate(todo):
            table[ti].p, table[ti].g, table[ti].buf = ptr(p), ptr(g), ptr(b)
            table[ti].n, table[ti].wd, table[ti].first = p.numel(), float(wd), int(first)
        o_off = (tbytes + 7) // 8 * 8
        t_off = o_off + 8 * nc
        (C.c_int64 * nc).from_address(host.data_ptr() + o_off)[:] = chunks_o
        (C.c_int32 * nc).from_address(host.data_ptr() + t_off)[:] = chunks_t
        devbuf = host[:t_off + 4 * nc].to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._copied[slot] = ev
        base = devbuf.data_ptr()
        from . import ops
        timers = ops.KERNEL_TIMERS if (ops.KERNEL_TIMER_ONLY is None or "sgd" in ops.KERNEL_TIMER_ONLY) else None
        if timers is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call("ttdg_sgd_multi_tensor", base, base + t_off, base + o_off, nc, CHUNK, float(lr), float(mom), stream())
        # the kernel writes through raw pointers: tell autograd (and every cache keyed on ``_version``, e.g. the folded
        # FrozenBN filters of modeling/backbone.py that the Dice pass reuses) that the parameters changed
        torch.autograd.graph.increment_version([t[0] for t in todo] + [t[2] for t in todo])
        if timers is not None:
            e1.record()
            nbytes = sum(p.numel() * 4 * (4 if first else 5) for p, _, _, _, first in todo)   # p r/w, g r, buf (r)/w
            timers.append(("sgd", e0, e1, nbytes, None))
        self._keepalive = (devbuf, [t[1] for t in todo])
        return None
