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
     run-to-run noise) in a few large buckets: ring all-reduce on xGMI is per-link bound (~153 GB/s/link), ~105 MB of fp32
     gradients = ~1.5 ms.  ``OverlappedGradReducer`` launches every bucket from autograd hooks as soon as its gradients
     exist (reverse layer order), so the all-reduce hides behind the rest of the backbone backward; ``allreduce_grads`` is
     the post-hoc form (same buckets, bit-identical results);
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


def plan_buckets(params, bucket_bytes=BUCKET_BYTES):
    """Static bucket layout shared by the post-hoc and the overlapped reduction: parameters in REVERSE registration order
    (the order their gradients become ready in a backward pass: heads / FPN first, res3 last), cut every ``bucket_bytes``.
    -> list of lists of indices into ``params``."""
    buckets, cur, nbytes = [], [], 0
    for i in reversed(range(len(params))):
        cur.append(i)
        nbytes += params[i].numel() * params[i].element_size()
        if nbytes >= bucket_bytes:
            buckets.append(cur)
            cur, nbytes = [], 0
    if cur:
        buckets.append(cur)
    return buckets


def _have_mask(params):
    """Which parameters have a gradient on ANY rank (one small MAX all-reduce)."""
    have = torch.tensor([p.grad is not None for p in params], dtype=torch.int32, device=params[0].device)
    _all_reduce_async(have, dist.ReduceOp.MAX).wait()
    return [bool(h) for h in have.tolist()]


def _is_cl(p):
    """A dense channels-last 4-D tensor (the detector's filters on the GPU, modeling/backbone.py) that is not ALSO NCHW-contiguous."""
    return p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last)


def storage_flat(g, like):
    """``g`` as a flat vector in the STORAGE order of the parameter ``like`` - a view when g has that layout (the usual case: no
    transposing copy of a channels-last gradient through reshape(-1)); the order depends on the parameter only, so every rank
    composes the same flat buffer."""
    return g.permute(0, 2, 3, 1).reshape(-1) if _is_cl(like) else g.reshape(-1)


def storage_unflat(flat, like):
    """Inverse of storage_flat: a tensor of ``like``'s shape AND strides over ``flat``'s memory (the fused SGD step then finds the
    gradient in the parameter's own layout and copies nothing)."""
    if _is_cl(like):
        n, c, h, w = like.shape
        return flat.view(n, h, w, c).permute(0, 3, 1, 2)
    return flat.view_as(like)


def _launch_bucket(params, idx, have, nsum, world):
    """Flatten the gradients of one bucket (zeros where this rank has none, replicated ones pre-divided) and start its
    SUM all-reduce.  -> (work handle, flat buffer, [(param, numel)]) or None for a bucket without any live gradient."""
    items = []
    for i in idx:
        if not have[i]:
            continue
        p = params[i]
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        items.append((p, g / world if i >= nsum else g))
    if not items:
        return None
    flat = torch.cat([storage_flat(g, p) for p, g in items])
    return _all_reduce_async(flat, dist.ReduceOp.SUM), flat, [(p, g.numel()) for p, g in items]


def _finish(work):
    for w, flat, items in work:
        w.wait()
        off = 0
        for p, n in items:
            p.grad = storage_unflat(flat[off:off + n], p)
            off += n


def allreduce_grads(summed, replicated, bucket_bytes=BUCKET_BYTES):
    """Post-hoc reduction (after ``loss.backward()`` has returned).  ``summed``: parameters whose gradients are partial sums
    over this rank's images (SUM over ranks); ``replicated``: parameters whose gradients are already complete on every rank
    (averaged).  A parameter that has no gradient on ANY rank keeps ``grad is None`` (the optimizer skips it, as the
    single-GPU step does); one that has a gradient somewhere gets zeros elsewhere."""
    params = list(summed) + list(replicated)
    if not params:
        return
    world = dist.get_world_size()
    have = _have_mask(params)
    work = [w for w in (_launch_bucket(params, idx, have, len(summed), world) for idx in plan_buckets(params, bucket_bytes)) if w is not None]
    _finish(work)


class OverlappedGradReducer:
    """The same reduction, launched DURING the backward pass ("bucketed to overlap with bwd").

    Every parameter carries a post-accumulate-grad hook; a bucket of ``plan_buckets`` is flattened and its all-reduce
    started as soon as all of its live parameters have their gradient AND every earlier bucket has been launched (the
    collectives must be issued in the same order on every rank), so the ring all-reduce of the FPN / res5 gradients is on
    the wire while res4 and res3 are still being differentiated.  ``finalize()`` (after backward) launches what is left -
    on a rank that held no graph no hook ever fires and everything is launched there, with zeros - waits, and installs the
    reduced gradients.

    Which parameters carry a gradient is DATA dependent (an FPN output convolution is only differentiated when some node of
    the step was sampled from its level), so the live set cannot be fixed once: it is the union of what has been seen so far
    (learned post-hoc in the first step), and every step ends with one small MAX all-reduce of "has a gradient on this rank"
    that settles the two remaining cases exactly as the post-hoc path does - a live parameter that received no gradient on
    ANY rank in this step goes back to ``grad is None`` (the optimizer skips it, as the single-GPU step does; its zeros were
    reduced for nothing), and a parameter outside the live set that did receive one is reduced in an extra bucket at the end
    and joins the live set.  Same values as the post-hoc path: bit-identical at world size 2 (tests/test_distributed.py); for
    more ranks equal up to the ring's reduction order, since the flat buffers are composed differently."""

    def __init__(self, summed, replicated, bucket_bytes=BUCKET_BYTES):
        self.params = list(summed) + list(replicated)
        self.nsum = len(summed)
        self.bucket_bytes = bucket_bytes
        self.buckets = plan_buckets(self.params, bucket_bytes)
        self.have = None                         # the live set, learned on the first step and grown afterwards
        self.overlapped_launches = 0             # buckets launched from inside backward (diagnostics / tests)
        self.late_joins = 0                      # parameters that joined the live set after the first step
        self._armed = False
        self._bucket_of = {}
        for b, idx in enumerate(self.buckets):
            for i in idx:
                self._bucket_of[i] = b
        self._handles = [p.register_post_accumulate_grad_hook(self._hook(i)) for i, p in enumerate(self.params)]

    def _hook(self, i):
        def fn(_p):
            if not self._armed or not self.have[i]:
                return                           # not armed, or a late joiner: settled in finalize()
            b = self._bucket_of[i]
            self._pending[b] -= 1
            self._advance(from_hook=True)
        return fn

    def _advance(self, from_hook):
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            w = _launch_bucket(self.params, self.buckets[self._next], self.have, self.nsum, self._world)
            if w is not None:
                self._work.append(w)
                self.overlapped_launches += int(from_hook)
            self._next += 1

    def prepare(self):
        """Call before ``loss.backward()`` (gradients must be None / zeroed: the hooks see accumulated gradients)."""
        self._world = dist.get_world_size()
        self._work, self._next = [], 0
        if self.have is None:
            self._armed = False
            return
        self._pending = [sum(1 for i in idx if self.have[i]) for idx in self.buckets]
        self._armed = True
        self._advance(from_hook=False)           # leading buckets without any live parameter

    def finalize(self):
        """Call after ``loss.backward()``."""
        if self.have is None:                    # first step: learn the live set, reduce post-hoc
            self.have = _have_mask(self.params)
            work = [w for w in (_launch_bucket(self.params, idx, self.have, self.nsum, self._world) for idx in self.buckets) if w is not None]
            _finish(work)
            return
        self._armed = False
        local = [p.grad is not None for p in self.params]    # before anything is installed
        for b in range(self._next, len(self.buckets)):       # not ready by hooks (this rank had no graph / no gradient)
            self._pending[b] = 0
        self._advance(from_hook=False)
        # every rank has now issued the same sequence of bucket collectives (some from inside backward, the rest just now);
        # only then the mask exchange - a rank without graphs would otherwise issue it BEFORE its buckets and the ranks' collective
        # orders would differ
        have = torch.tensor(local, dtype=torch.int32, device=self.params[0].device)
        _all_reduce_async(have, dist.ReduceOp.MAX).wait()
        anywhere = [bool(h) for h in have.tolist()]          # gradients that exist on SOME rank in this step
        late = [i for i, (a, h) in enumerate(zip(anywhere, self.have)) if a and not h]
        if late:                                 # outside the live set so far: one extra bucket, the same on every rank
            w = _launch_bucket(self.params, late, anywhere, self.nsum, self._world)
            if w is not None:
                self._work.append(w)
            self.late_joins += len(late)
        _finish(self._work)
        self._work = []
        for i, (a, h) in enumerate(zip(anywhere, self.have)):
            if h and not a:
                self.params[i].grad = None       # no gradient anywhere in this step: skipped by the optimizer, as post-hoc
            elif a and not h:
                self.have[i] = True

    def remove(self):
        for h in self._handles:
            h.remove()


def split_params(model):
    """(summed, replicated) for a DAobjTwoStagePseudoLabGeneralizedRCNN: the matching modules see the whole gathered
    multi-graph on every rank, everything else sees only the local images."""
    summed, replicated = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (replicated if name.startswith(("multi_matching_unsup.", "multi_matching_sup.")) else summed).append(p)
    return summed, replicated
