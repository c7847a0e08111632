"""A/B of the backbone's 1 x 1 convolutions inside the network: ResNet-50-FPN at the bench shape
(4 x 3 x 800 x 800, fp32, channels-last, stem + res2 frozen), TTA-style forward + backward and the no-grad forward, with
modeling.backbone.OWN_POINTWISE on (fused streaming product, csrc/pointwise.hip) and off (MIOpen + one-pass epilogue kernel).
Per arm: wall time per step and the HOST time to enqueue one step (a step whose enqueue time equals its wall time is host-bound).
usage: ab_pointwise.py [reps]"""
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
    dev = torch.device("cuda:0")
    cfg = get_cfg()
    cfg.MODEL.DEVICE = "cuda:0"
    torch.manual_seed(0)
    model = build_model(cfg)
    net = model.backbone
    x = torch.randn(4, 3, 800, 800, device=dev)

    def step(mode):
        if mode == "train":
            out = net(x)
            loss = sum(v.float().square().mean() for v in out.values())
            for p in net.parameters():
                p.grad = None
            loss.backward()
        else:
            with torch.no_grad():
                net(x)

    for own in (True, False, True, False):
        bb.OWN_POINTWISE = own
        for mode in ("train", "eval"):
            net.train(mode == "train")
            for _ in range(3):
                step(mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            enq = 0.0
            for _ in range(reps):
                e0 = time.perf_counter()
                step(mode)
                enq += time.perf_counter() - e0
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e3
            # host-only: the same step enqueued while the GPU is still busy with a long queue never waits for the GPU except at syncs
            print("%-28s %-5s wall %7.2f ms   enqueue %7.2f ms" % ("own pointwise" if own else "vendor + epilogue kernel", mode, wall, enq / reps * 1e3), flush=True)


def per_block():
    """no-grad forward, HIP events around every bottleneck: which blocks gain from the fused product, in situ"""
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.modeling import backbone as bb
    from ttdg_mgm_amd.modeling import build_model
    dev = torch.device("cuda:0")
    cfg = get_cfg()
    cfg.MODEL.DEVICE = "cuda:0"
    torch.manual_seed(0)
    net = build_model(cfg).backbone.eval()
    x = torch.randn(4, 3, 800, 800, device=dev)
    blocks = [(n, m) for n, m in net.named_modules() if isinstance(m, bb.Bottleneck)]
    rec = {}
    cur = {}

    def pre(name):
        def f(mod, inp):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            cur[name] = e
        return f

    def post(name):
        def f(mod, inp, out):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            rec.setdefault((name, bb.OWN_POINTWISE), []).append((cur[name], e))
        return f
    for n, m in blocks:
        m.register_forward_pre_hook(pre(n))
        m.register_forward_hook(post(n))
    for own in (True, False) * 6:
        bb.OWN_POINTWISE = own
        with torch.no_grad():
            net(x)
    torch.cuda.synchronize()
    tot = [0.0, 0.0]
    for n, _ in blocks:
        t = [sorted(a.elapsed_time(b) * 1e3 for a, b in rec[(n, own)][2:]) for own in (True, False)]
        med = [v[len(v) // 2] for v in t]
        tot[0] += med[0]
        tot[1] += med[1]
        print("%-24s own %7.1f us   vendor + epilogue %7.1f us   x%.2f" % (n, med[0], med[1], med[1] / med[0]), flush=True)
    print("all bottlenecks: own %.1f us, vendor + epilogue %.1f us" % tuple(tot))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "blocks":
        per_block()
        sys.exit(0)
    main()
