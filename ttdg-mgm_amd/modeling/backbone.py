"""ResNet-50 + FPN (p2..p6, 256 ch) with FrozenBN; stem and res2 frozen (detectron2 defaults [3P], SURVEY.md App. C:
FREEZE_AT 2, NORM FrozenBN, STRIDE_IN_1X1 True, FPN top block LastLevelMaxPool).  Module/parameter names follow
detectron2 (``bottom_up.stem.conv1.norm.weight``, ``fpn_lateral2`` ...)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


# The four switches below are module attributes, not configuration: nothing reads the environment.  The product runs with all of
# them True; the parity tests flip one at a time to compare a fused kernel with the plain torch formulation of the same arithmetic
# (tests/test_gpu_parity.py), tools/ab_backbone.py measures them, and calibrate_frozen_bn turns FUSED_EPILOGUE off while its
# statistics hooks sit on ConvNorm.forward.
FUSED_EPILOGUE = True
# the FrozenBN scale is folded into all filters of a ResNet stage by ONE launch (ops.row_scale_multi) instead of one elementwise
# kernel per filter per pass
MULTI_FOLD = True
# bias (+ top-down sum) of the FPN convolutions, the RPN head inside the TTA step and the mask head through the in-place epilogue kernel
FUSED_HEADS = True
# The backbone runs in CHANNELS-LAST memory on the GPU (fp32, outside autocast): MIOpen's fastest fp32 kernels on gfx950 are its
# NHWC implicit GEMMs, which it wraps in transposes when handed NCHW tensors (6 % of an adapted batch); measured with MIOpen
# choosing per shape in both layouts: 21.6 vs 24.2 ms per train step, 8.5 vs 9.5 ms per no-grad forward (plain epilogues).
# FPN.forward moves its filters to channels-last storage at the first GPU forward and converts the image batch; every kernel behind ops.* takes either
# layout.  (False keeps NCHW: the layout parity test.)
CHANNELS_LAST = True
# The pointwise (1 x 1) convolutions of the channels-last backbone - bottleneck conv1 / conv3 / projection shortcut, FPN laterals -
# run on the repo's own streaming MFMA product with shift, residual (+ the FPN's up-sampled top-down map) and ReLU applied in its
# epilogue (ops.pointwise_conv, csrc/pointwise.hip): no epilogue pass over HBM, and where gradients flow the two backward products
# without MIOpen's atomic weight-gradient kernels and their zero fills (ops.PointwiseConvFn).  False: vendor convolution + the
# one-pass epilogue kernel (the A/B arm, and the exact-equality parity tests of the epilogue kernels).
OWN_POINTWISE = True
# ... and in the blocks without a tape the 3 x 3 convolution's own epilogue (shift + ReLU) is applied by the NEXT product while it
# fetches its operand (ops.pointwise_conv(pbias=, prelu=)), instead of an in-place pass over the 3 x 3 output
FUSED_INPUT_ACTIVATION = True
# ... and the projection shortcut of a stage's first block is accumulated into conv3's product as a second reduction segment
FUSED_SHORTCUT = True
# stages whose pointwise convolutions stay on the vendor kernels: at 25 x 25 x 4 = 2500 pixels the streaming product's 64 x 64 tiles
# leave CUs idle (in-situ A/B per block, profiles/r06_pointwise_ab_in_situ.txt: res2 - res4 x1.04 - 1.18, res5 x0.91 - 0.96)
POINTWISE_MIN_PIXELS = 4096


_CAPTURING = False      # set by modeling/graphed.py while a hipGraph capture records the forward: no host synchronisation then


def _publish(t):
    """A lazily built constant is about to be cached and may be consumed from ANOTHER HIP stream (the Dice pass runs
    inference on several streams): make sure the kernels that produce it have finished before it becomes visible."""
    if t.is_cuda and not _CAPTURING:
        torch.cuda.current_stream(t.device).synchronize()
    return t


def _fusable(x, *mods):
    """True when no gradient can flow through this piece (frozen filters on a constant input, or no_grad): the fp32 GPU
    forward may then run in place with the fused shift / residual / ReLU kernel (ops.bias_act_)."""
    if not (FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32) or torch.is_autocast_enabled():
        return False              # (under autocast the convolution output is bf16: the fp32 kernel must not touch it)
    if not torch.is_grad_enabled():
        return True
    return not x.requires_grad and not any(p.requires_grad for m in mods for p in m.parameters())


def _fusable_on_tape(x):
    """The adapted blocks (gradients flow): fp32 GPU tensors outside autocast take the in-place epilogue WITH its backward
    (ops.BiasActFn); anything else (CPU host pipeline, bf16 autocast, calibration hooks) keeps the plain torch formulation."""
    return FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and torch.is_grad_enabled()


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c) - eps)

        self._fold = None

    def folded(self):
        """(scale (C,1,1,1), shift (C,)) of the frozen affine map; cached until a buffer is modified in place."""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               self.weight.device, self.weight.data_ptr())
        if self._fold is None or self._fold[0] != key:
            with torch.no_grad():
                scale = self.weight * (self.running_var + self.eps).rsqrt()
                shift = _publish(self.bias - self.running_mean * scale)
            self._fold = (key, scale.view(-1, 1, 1, 1), shift)
        return self._fold[1], self._fold[2]

    def forward(self, x):
        scale, shift = self.folded()
        return x * scale.view(1, -1, 1, 1).to(x.dtype) + shift.view(1, -1, 1, 1).to(x.dtype)


class ConvNorm(nn.Conv2d):
    """Conv2d carrying its FrozenBN as ``.norm`` (detectron2 Conv2d wrapper layout)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, norm=True):
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=not norm)
        self.norm = FrozenBatchNorm2d(cout) if norm else None
        self._wfold = None
        self._staged_w = None            # the folded filter of the current forward pass, when the stage folded all of its filters at once
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")

    def folded_const(self):
        """(folded filter, shift) as constants: frozen filter or a no-grad pass; cached until the weights move."""
        scale, shift = self.norm.folded()
        key = (self.weight._version, self.weight.data_ptr(), self.norm._fold[0])
        if self._wfold is None or self._wfold[0] != key:
            with torch.no_grad():
                self._wfold = (key, _publish(self.weight * scale))
        return self._wfold[1], shift

    def folded_on_tape(self):
        """(folded filter on the autograd tape, shift) for a filter that is being adapted."""
        scale, shift = self.norm.folded()
        return (self._staged_w if self._staged_w is not None else self.weight * scale), shift

    def raw(self, x):
        """(convolution with the folded filter, WITHOUT its shift; the shift) - for the fused in-place epilogue."""
        w, shift = self.folded_const()
        return F.conv2d(x, w, None, self.stride, self.padding), shift

    def raw_on_tape(self, x):
        """raw() for a filter that is being adapted: the folded filter stays on the autograd tape."""
        scale, shift = self.norm.folded()
        w = self._staged_w if self._staged_w is not None else self.weight * scale
        return F.conv2d(x, w, None, self.stride, self.padding), shift

    def forward(self, x):
        if self.norm is None:
            return super().forward(x)
        # frozen statistics: fold the per-channel scale into the filter and pass the shift as the conv bias
        # (same arithmetic as conv -> x*scale+shift, one activation-sized pass less in forward and backward)
        scale, shift = self.norm.folded()
        if self.weight.requires_grad and torch.is_grad_enabled():
            w = self.weight * scale                  # stays on the autograd tape: the filter is being adapted
        else:
            # frozen filter (stem / res2) or the eval pass: the folded filter is a constant until the weights move
            key = (self.weight._version, self.weight.data_ptr(), self.norm._fold[0])
            if self._wfold is None or self._wfold[0] != key:
                with torch.no_grad():
                    self._wfold = (key, _publish(self.weight * scale))
            w = self._wfold[1]
        return F.conv2d(x, w, shift.to(x.dtype), self.stride, self.padding)


def _fold_stage(stage, x):
    """Fold the FrozenBN scales into every filter of ``stage`` that needs it, in one launch, before its blocks run.
    Gradients flow (the adapted stages of a TTA step): the folded filters come from ops.FoldFiltersFn and are handed to the
    blocks through ``ConvNorm._staged_w``; returns the list to release after the stage.  No gradient (Dice pass, frozen stages):
    refreshes the stale entries of the per-filter caches ``ConvNorm.raw`` reads.  Same products as the per-filter multiplies."""
    if not (MULTI_FOLD and FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32) or torch.is_autocast_enabled():
        return None
    convs = stage.__dict__.get("_convnorms")
    if convs is None:
        convs = stage.__dict__["_convnorms"] = [m for m in stage.modules() if isinstance(m, ConvNorm) and m.norm is not None]
    if not convs:
        return None
    if torch.is_grad_enabled() and (x.requires_grad or any(c.weight.requires_grad for c in convs)):
        scales = tuple(c.norm.folded()[0].view(-1) for c in convs)
        for c, w in zip(convs, ops.FoldFiltersFn.apply(scales, *[c.weight for c in convs])):
            c._staged_w = w
        return convs
    stale = []
    for c in convs:
        scale = c.norm.folded()[0]
        key = (c.weight._version, c.weight.data_ptr(), c.norm._fold[0])
        if c._wfold is None or c._wfold[0] != key:
            stale.append((c, key, scale.view(-1)))
    if len(stale) > 1:
        with torch.no_grad():
            outs = ops.row_scale_multi([c.weight for c, _, _ in stale], [sc for _, _, sc in stale])
        _publish(outs[0])
        for (c, key, _), w in zip(stale, outs):
            c._wfold = (key, w)
    return None


class Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = ConvNorm(3, 64, 7, stride=2, padding=3)

    def forward(self, x):
        if _fusable(x, self):
            y, b = self.conv1.raw(x)
            return F.max_pool2d(ops.bias_act_(y, b), kernel_size=3, stride=2, padding=1)
        return F.max_pool2d(F.relu_(self.conv1(x)), kernel_size=3, stride=2, padding=1)


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, mid, stride):
        super().__init__()
        self.shortcut = ConvNorm(cin, cout, 1, stride=stride) if cin != cout else None
        self.conv1 = ConvNorm(cin, mid, 1, stride=stride)          # STRIDE_IN_1X1
        self.conv2 = ConvNorm(mid, mid, 3, padding=1)
        self.conv3 = ConvNorm(mid, cout, 1)

    def _own(self, x):
        st = self.conv1.stride[0]
        pixels = x.shape[0] * ((x.shape[2] - 1) // st + 1) * ((x.shape[3] - 1) // st + 1)
        return OWN_POINTWISE and pixels >= POINTWISE_MIN_PIXELS and ops.pointwise_ok(x, self.conv1.weight) and ops.pointwise_ok(x, self.conv3.weight)

    def forward(self, x):
        if _fusable(x, self) and self._own(x):
            # no gradient through this block: the two pointwise convolutions (and the projection shortcut) on the streaming product
            # with their epilogues fused, the 3 x 3 convolution on the vendor kernel + the in-place epilogue
            w1, b1 = self.conv1.folded_const()
            y = ops.pointwise_conv(x, w1, b1, relu=True, stride=self.conv1.stride[0])
            y, b2 = self.conv2.raw(y)
            w3, b3 = self.conv3.folded_const()
            if self.shortcut is not None:
                ws, bs = self.shortcut.folded_const()
                if FUSED_SHORTCUT and FUSED_INPUT_ACTIVATION and x.shape[1] % 32 == 0 and y.shape[1] % 32 == 0:
                    # the projection shortcut is a second reduction segment of conv3's product: its output is never written or re-read
                    return ops.pointwise_conv(y, w3, b3, bias2=bs, relu=True, pbias=b2, prelu=True, second=(x, ws, self.shortcut.stride[0]))
                sc = ops.pointwise_conv(x, ws, bs, relu=False, stride=self.shortcut.stride[0])
            else:
                sc = x
            if FUSED_INPUT_ACTIVATION:
                # conv2's shift + ReLU ride on conv3's operand fetch (relu(y + b2[k]) on the fragments): the 3 x 3 output is read once
                return ops.pointwise_conv(y, w3, b3, residual=sc, relu=True, pbias=b2, prelu=True)
            return ops.pointwise_conv(ops.bias_act_(y, b2), w3, b3, residual=sc, relu=True)
        if _fusable_on_tape(x) and self._own(x):
            pw, fused = ops.PointwiseConvFn.apply, ops.BiasActFn.apply
            w1, b1 = self.conv1.folded_on_tape()
            st = self.conv1.stride[0]
            if st > 1:                        # conv1 and the projection shortcut read the same strided pixels: compacted once
                x = ops.StridedSliceFn.apply(x, st)
            y = pw(x, w1, b1, None, None, True, 1, False)
            y, b2 = self.conv2.raw_on_tape(y)
            y = fused(y, b2, None, None)
            w3, b3 = self.conv3.folded_on_tape()
            if self.shortcut is not None:
                ws, bs = self.shortcut.folded_on_tape()
                sc = pw(x, ws, bs, None, None, False, 1, False)
            else:
                sc = x
            return pw(y, w3, b3, sc, None, True, 1, False)
        if _fusable(x, self):
            # no gradient through this block: three in-place epilogues instead of eight elementwise passes
            y, b = self.conv1.raw(x)
            y, b = self.conv2.raw(ops.bias_act_(y, b))
            y, b = self.conv3.raw(ops.bias_act_(y, b))
            if self.shortcut is not None:
                sc, bs = self.shortcut.raw(x)
                return ops.bias_act_(y, b, sc, bs)
            return ops.bias_act_(y, b, ops.like_layout(x, y))
        if _fusable_on_tape(x):
            # gradients flow: the same three in-place epilogues, each paired with a one-pass backward (ops.BiasActFn)
            fused = ops.BiasActFn.apply
            y, b = self.conv1.raw_on_tape(x)
            y, b = self.conv2.raw_on_tape(fused(y, b, None, None))
            y, b = self.conv3.raw_on_tape(fused(y, b, None, None))
            if self.shortcut is not None:
                sc, bs = self.shortcut.raw_on_tape(x)
                return fused(y, b, sc, bs)
            return fused(y, b, ops.like_layout(x, y), None)
        out = F.relu_(self.conv1(x))
        out = F.relu_(self.conv2(out))
        out = self.conv3(out)
        return F.relu_(out + (self.shortcut(x) if self.shortcut is not None else x))


class ResNet50(nn.Module):
    def __init__(self, freeze_at=2):
        super().__init__()
        self.stem = Stem()
        cfgs = [("res2", 3, 64, 256, 64, 1), ("res3", 4, 256, 512, 128, 2), ("res4", 6, 512, 1024, 256, 2), ("res5", 3, 1024, 2048, 512, 2)]
        for name, n, cin, cout, mid, stride in cfgs:
            blocks = [Bottleneck(cin if i == 0 else cout, cout, mid, stride if i == 0 else 1) for i in range(n)]
            setattr(self, name, nn.Sequential(*blocks))
        frozen = [self.stem] + [getattr(self, "res%d" % i) for i in range(2, freeze_at + 1)]
        for m in frozen[:freeze_at]:
            for p in m.parameters():
                p.requires_grad_(False)

    # cfg-5 precision islands (rcnn.autocast_backbone = "res2" ... "res5"): the stages up to and including ``autocast_upto`` run
    # under bf16 autocast, the later ones (and the FPN) in fp32 on the up-cast activations; None = whatever context the caller set
    autocast_upto = None
    _ORDER = ("stem", "res2", "res3", "res4", "res5")

    def _in_island(self, name):
        return self._ORDER.index(name) <= self._ORDER.index(self.autocast_upto)

    def _run(self, name, mod, x):
        """``x`` is already fp32 for a stage behind the island (forward up-casts ONCE per stage boundary)."""
        if self.autocast_upto is None:
            return mod(x)
        if self._in_island(name):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return mod(x)
        return mod(x)

    def forward(self, x):
        x = self._run("stem", self.stem, x)
        outs = []
        for name in ("res2", "res3", "res4", "res5"):
            stage = getattr(self, name)
            islands = self.autocast_upto is not None
            island = islands and self._in_island(name)
            if islands and not island and x.dtype != torch.float32:
                x = x.float()                      # the one up-cast at the island's edge: fold, stage and feature map share it
            staged = None if island else _fold_stage(stage, x)
            try:
                x = self._run(name, stage, x)
            finally:
                for c in staged or ():
                    c._staged_w = None
            outs.append(x.float() if islands and x.dtype != torch.float32 else x)
        return tuple(outs)


class FPN(nn.Module):
    size_divisibility = 32
    out_channels = 256
    strides = (4, 8, 16, 32, 64)
    _out_feature_channels = {"p2": 256, "p3": 256, "p4": 256, "p5": 256, "p6": 256, "res4": 1024, "vgg4": 512}

    def __init__(self, freeze_at=2):
        super().__init__()
        self.bottom_up = ResNet50(freeze_at)
        for i, c in zip((2, 3, 4, 5), (256, 512, 1024, 2048)):
            lat, out = nn.Conv2d(c, 256, 1), nn.Conv2d(256, 256, 3, padding=1)
            for m in (lat, out):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)
            setattr(self, "fpn_lateral%d" % i, lat)
            setattr(self, "fpn_output%d" % i, out)

    @staticmethod
    def _conv(conv, x, residual=None):
        """conv(x) + bias (+ residual).  fp32 on the GPU: the bias (and the top-down sum) is applied in place by the vectorised
        epilogue kernel - with its backward when gradients flow - instead of torch's broadcast add_ (3.3 TB/s on p2) and add."""
        if not (FUSED_HEADS and FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32) or torch.is_autocast_enabled():
            y = conv(x)
            return y if residual is None else y + residual
        y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding)
        if torch.is_grad_enabled() and (y.requires_grad or conv.bias.requires_grad or (residual is not None and residual.requires_grad)):
            return ops.BiasAddFn.apply(y, conv.bias, residual)
        return ops.bias_act_(y, conv.bias, residual, None, relu=False)

    @staticmethod
    def _lateral(conv, c, coarse=None):
        """lateral(c) + bias (+ the coarser level, nearest-neighbour up-sampled): one fused product on the GPU's channels-last fp32
        path (the up-sampled map is never materialised), conv + interpolate + add otherwise."""
        if OWN_POINTWISE and FUSED_HEADS and FUSED_EPILOGUE and not torch.is_autocast_enabled() and ops.pointwise_ok(c, conv.weight) \
                and c.shape[0] * c.shape[2] * c.shape[3] >= POINTWISE_MIN_PIXELS \
                and (coarse is None or (coarse.is_contiguous(memory_format=torch.channels_last) and coarse.dtype == torch.float32
                                        and coarse.shape[2] * 2 == c.shape[2] and coarse.shape[3] * 2 == c.shape[3])):
            if torch.is_grad_enabled() and (c.requires_grad or conv.weight.requires_grad or conv.bias.requires_grad or (coarse is not None and coarse.requires_grad)):
                return ops.PointwiseConvFn.apply(c, conv.weight, conv.bias, coarse, None, False, 1, coarse is not None)
            return ops.pointwise_conv(c, conv.weight, conv.bias, residual=coarse, res_up=coarse is not None)
        return FPN._conv(conv, c, None if coarse is None else F.interpolate(coarse, scale_factor=2.0, mode="nearest"))

    def forward(self, x):
        if CHANNELS_LAST and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
            if not self.fpn_output2.weight.is_contiguous(memory_format=torch.channels_last):
                # first GPU forward: the filters move to (O, kh, kw, I) storage, in place (same Parameter objects: optimizers, hooks
                # and state dicts are unaffected; the fused SGD step works on either dense layout).  The CPU model keeps NCHW.
                self.to(memory_format=torch.channels_last)
            x = x.contiguous(memory_format=torch.channels_last)
        c2, c3, c4, c5 = self.bottom_up(x)
        prev = self._lateral(self.fpn_lateral5, c5)
        p5 = self._conv(self.fpn_output5, prev)
        outs = [p5]
        for i, c in ((4, c4), (3, c3), (2, c2)):
            prev = self._lateral(getattr(self, "fpn_lateral%d" % i), c, prev)
            outs.insert(0, self._conv(getattr(self, "fpn_output%d" % i), prev))
        outs.append(F.max_pool2d(p5, kernel_size=1, stride=2, padding=0))
        return dict(zip(("p2", "p3", "p4", "p5", "p6"), outs))
