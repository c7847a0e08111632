"""Mode S — one synchronous "universe graph" over all ranks (BASELINE.json north_star: "RCCL all-gather
of the universe graph over xGMI").

Mode R (the default, and what the reference does) gives every rank its own shard and its own adaptation trajectory.  Mode S
makes the N ranks take ONE adaptation step on the N*B images they hold together, i.e. the single-GPU algorithm at batch
N*B with the backbone work sharded:

  1. every rank runs the backbone and the node sampler on its B images;
  2. ONE all-gather of the node embeddings (+ labels) — <= 64 graphs x ~35 nodes x 257 fp32 = ~2 MB, latency-bound on
     xGMI — after which every rank holds the global multi-graph (a second, tiny all-gather carries the node counts);
  3. the matching step (affinity, Sinkhorn, GA-MGM, loss) is REPLICATED: it is ~3 % of the step's FLOPs, its inputs are
     bit-identical on every rank after the all-gather and the kernels are deterministic, so replicating it costs less
     than partitioning the pair blocks and exchanging Wds (an extra all-gather of M x M floats plus a reduce-scatter of
     node gradients on the critical path);
  4. backward: each rank keeps its own rows of d loss / d nodes (no reduce-scatter needed: the replicated loss already
     holds every pair's contribution) and back-propagates through its own backbone activations;
  5. gradient all-reduce (SUM over ranks for backbone / FPN tensors, whose gradients are partial sums over images; the
     matching module's own gradients are already complete on every rank and are averaged, which only washes out
     run-to-run noise) in a few large buckets launched back to back: ring all-reduce on xGMI is per-link bound
     (~153 GB/s/link), ~105 MB of fp32 gradients = ~1.5 ms, and bucket k+1 is packed while bucket k is on the wire;
  6. the fused SGD step, identical on every rank, so the replicas never drift.

Ranks whose shard has run out of batches keep taking part with zero graphs (``inputs=None``).
"""
import torch
import torch.distributed as dist

MAX_GRAPHS = 64                 # include/ttdg_mgm.h TTDG_MAX_GRAPHS: the global multi-graph must fit one descriptor
BUCKET_BYTES = 32 << 20


def enabled():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _staged(t):
    """gloo (CPU tests, validation on a one-GPU box) has no device all-gather: stage device tensors through the host.
    RCCL ("nccl") takes device buffers directly."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_gather(t):
    world = dist.get_world_size()
    src = t.cpu() if _staged(t) else t
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    return [p.to(t.device) for p in parts] if _staged(t) else parts


class _Done:
    def wait(self):
        pass


def _all_reduce_async(t, op):
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
        return _Done()
    return dist.all_reduce(t, op=op, async_op=True)


class _GatherGraphs(torch.autograd.Function):
    """(m_r, d) local node rows + labels -> the rows of every rank in rank order.  Backward: this rank's rows of the
    incoming gradient (the loss is replicated, so no cross-rank reduction is due here)."""

    @staticmethod
    def forward(ctx, x, labels, counts, rank):
        d = x.shape[1]
        mmax = max(max(counts), 1)
        pad = x.new_zeros(mmax, d + 1)
        pad[:x.shape[0], :d] = x
        pad[:x.shape[0], d] = labels.to(x.dtype)
        parts = _all_gather(pad)
        rows = torch.cat([p[:n] for p, n in zip(parts, counts)], dim=0)
        ctx.lo, ctx.n = sum(counts[:rank]), counts[rank]
        lab = rows[:, d].round().long()
        ctx.mark_non_differentiable(lab)
        return rows[:, :d].contiguous(), lab

    @staticmethod
    def backward(ctx, g, _):
        return g[ctx.lo:ctx.lo + ctx.n], None, None, None


def gather_graphs(nodes, labels, device, dim=256):
    """nodes / labels: this rank's lists (or None when it has no graph).  Returns the global lists, rank-major."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [] if nodes is None else [int(x.shape[0]) for x in nodes]
    if len(sizes) > MAX_GRAPHS:
        raise ValueError("more than {} graphs on one rank".format(MAX_GRAPHS))
    meta = torch.zeros(MAX_GRAPHS + 1, dtype=torch.int64, device=device)
    meta[0] = len(sizes)
    if sizes:
        meta[1:1 + len(sizes)] = torch.tensor(sizes, dtype=torch.int64)
    metas = torch.stack(_all_gather(meta)).tolist()                  # one host read: the graph sizes drive every launch downstream
    all_sizes = [m[1:1 + m[0]] for m in metas]
    counts = [sum(s) for s in all_sizes]
    if sum(len(s) for s in all_sizes) > MAX_GRAPHS:
        raise ValueError("the global multi-graph holds more than {} graphs".format(MAX_GRAPHS))
    if sizes:
        x, lab = torch.cat(nodes, dim=0), torch.cat([l.reshape(-1) for l in labels])
    else:
        x, lab = torch.zeros(0, dim, device=device), torch.zeros(0, dtype=torch.int64, device=device)
    rows, lab_all = _GatherGraphs.apply(x.float(), lab, counts, rank)
    flat = [n for s in all_sizes for n in s]
    if not flat:
        return None, None
    return list(torch.split(rows, flat)), list(torch.split(lab_all, flat))


def allreduce_grads(summed, replicated, bucket_bytes=BUCKET_BYTES):
    """``summed``: parameters whose gradients are partial sums over this rank's images (SUM over ranks);
    ``replicated``: parameters whose gradients are already complete on every rank (averaged).  A parameter that has no
    gradient on ANY rank keeps ``grad is None`` (the optimizer skips it, as the single-GPU step does); one that has a
    gradient somewhere gets zeros elsewhere."""
    params = list(summed) + list(replicated)
    if not params:
        return
    world = dist.get_world_size()
    dev = params[0].device
    have = torch.tensor([p.grad is not None for p in params], dtype=torch.int32, device=dev)
    _all_reduce_async(have, dist.ReduceOp.MAX).wait()
    have = have.tolist()
    nsum = len(summed)
    work, bucket, nbytes = [], [], 0

    def flush():
        nonlocal bucket, nbytes
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for _, g in bucket])
        work.append((_all_reduce_async(flat, dist.ReduceOp.SUM), flat, bucket))
        bucket, nbytes = [], 0

    for i, p in enumerate(params):
        if not have[i]:
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if i >= nsum:
            g = g / world
        bucket.append((p, g))
        nbytes += g.numel() * g.element_size()
        if nbytes >= bucket_bytes:
            flush()
    flush()
    for w, flat, items in work:
        w.wait()
        off = 0
        for p, g in items:
            n = g.numel()
            p.grad = flat[off:off + n].view_as(p)
            off += n


def split_params(model):
    """(summed, replicated) for a DAobjTwoStagePseudoLabGeneralizedRCNN: the matching modules see the whole gathered
    multi-graph on every rank, everything else sees only the local images."""
    summed, replicated = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (replicated if name.startswith(("multi_matching_unsup.", "multi_matching_sup.")) else summed).append(p)
    return summed, replicated
