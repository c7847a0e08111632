"""A/B of the vendor-kernel side of a TTA step: ResNet-50-FPN forward + backward at the bench's shape
(4 x 3 x 800 x 800, fp32, stem + res2 frozen) with (a) the product's NCHW path + fused epilogues, (b) NCHW with plain torch
element-wise ops, (c) channels_last tensors end to end (plain ops), and the eval-mode forward of each.  MIOpen's find mode is
selected from outside (MIOPEN_FIND_MODE=...).  usage: ab_backbone.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.modeling import backbone as bb
    from ttdg_mgm_amd.modeling import build_model
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    if os.environ.get("AB_BENCHMARK") == "1":          # PyTorch then lets MIOpen time its solvers per shape (first use of a shape: seconds)
        torch.backends.cudnn.benchmark = True
    only = os.environ.get("AB_ONLY", "")
    dev = torch.device("cuda:0")
    cfg = get_cfg()
    cfg.MODEL.DEVICE = "cuda:0"
    torch.manual_seed(0)
    model = build_model(cfg)
    net = model.backbone
    x = torch.randn(4, 3, 800, 800, device=dev)
    print("MIOPEN_FIND_MODE =", os.environ.get("MIOPEN_FIND_MODE"), " MIOPEN_FIND_ENFORCE =", os.environ.get("MIOPEN_FIND_ENFORCE"))

    def run(label, fused, cl):
        bb.FUSED_EPILOGUE = fused
        n = net.to(memory_format=torch.channels_last) if cl else net.to(memory_format=torch.contiguous_format)
        xi = x.contiguous(memory_format=torch.channels_last) if cl else x
        for mode in ("train", "eval"):
            n.train(mode == "train")

            def step():
                if mode == "train":
                    out = n(xi)
                    loss = sum(v.float().square().mean() for v in out.values())
                    for p in n.parameters():
                        p.grad = None
                    loss.backward()
                else:
                    with torch.no_grad():
                        n(xi)
            tw = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            tw = time.perf_counter() - tw
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            print("%-34s %-5s %8.2f ms   (3 warm-up steps took %.1f s; cudnn.benchmark %s)" % (label, mode, (time.perf_counter() - t0) / reps * 1e3, tw,
                  torch.backends.cudnn.benchmark), flush=True)

    if only == "miopen_fused":
        # experiment: MIOpen's own conv + bias (+ residual) + ReLU fusion plans (aten::miopen_convolution_relu / _add_relu) for the
        # no-grad forward, instead of convolution + our one-pass epilogue
        import torch.nn.functional as F

        def folded(c):
            scale, shift = c.norm.folded()
            return (c.weight * scale).detach(), shift

        def conv_relu(c, x):
            w, b = folded(c)
            return torch.miopen_convolution_relu(x, w, b, c.stride, c.padding, (1, 1), 1)

        def block(self, x):
            y = conv_relu(self.conv1, x)
            y = conv_relu(self.conv2, y)
            w3, b3 = folded(self.conv3)
            if self.shortcut is not None:
                ws, bs = folded(self.shortcut)
                z = F.conv2d(x, ws, bs, self.shortcut.stride, self.shortcut.padding)
            else:
                z = x
            return torch.miopen_convolution_add_relu(y, w3, z, 1.0, b3, self.conv3.stride, self.conv3.padding, (1, 1), 1)

        def stem(self, x):
            return F.max_pool2d(conv_relu(self.conv1, x), kernel_size=3, stride=2, padding=1)
        net.eval()
        with torch.no_grad():
            ref = {k: v.clone() for k, v in net(x).items()}
        bb.Bottleneck.forward, bb.Stem.forward = block, stem
        with torch.no_grad():
            got = net(x)
            print("max |fused - product| relative:", max(float((got[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ref))
            for _ in range(3):
                net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                net(x)
            torch.cuda.synchronize()
        print("%-34s %-5s %8.2f ms" % ("MIOpen fusion plans (conv+bias+relu)", "eval", (time.perf_counter() - t0) / reps * 1e3), flush=True)
        return
    if only == "fpn_cl":
        # experiment: only the FPN 3x3 output convolutions (the igemm NHWC kernels + MIOpen's transposes around them) see
        # channels_last tensors; the trunk stays NCHW
        import torch.nn.functional as F
        CL = torch.channels_last

        def fpn_forward(self, x):
            c2, c3, c4, c5 = self.bottom_up(x)
            prev = self.fpn_lateral5(c5)
            outs = [self.fpn_output5(prev.contiguous(memory_format=CL))]
            for i, c in ((4, c4), (3, c3), (2, c2)):
                prev = getattr(self, "fpn_lateral%d" % i)(c) + F.interpolate(prev, scale_factor=2.0, mode="nearest")
                outs.insert(0, getattr(self, "fpn_output%d" % i)(prev.contiguous(memory_format=CL)))
            outs.append(F.max_pool2d(outs[-1], kernel_size=1, stride=2, padding=0))
            return dict(zip(("p2", "p3", "p4", "p5", "p6"), outs))
        for i in (2, 3, 4, 5):
            m = getattr(net, "fpn_output%d" % i)
            m.weight.data = m.weight.data.contiguous(memory_format=CL)
        type(net).forward = fpn_forward
        run("FPN output convs channels_last", True, False)
        return
    if only in ("", "nchw"):
        run("NCHW + fused epilogues (product)", True, False)
        run("NCHW, plain torch epilogues", False, False)
    if only in ("", "cl"):
        run("channels_last, plain epilogues", False, True)


if __name__ == "__main__":
    main()
