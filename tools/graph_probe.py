"""Diagnostics (not product): why did the hipGraph replay of the TTA step's backbone backward return stale gradients (VERDICT r4 item 6,
modeling/graphed.py)?  The backbone's forward + backward at the bench shape (4 x 3 x 800 x 800, channels-last, fp32) once eagerly and
once through torch.cuda.make_graphed_callables, for TWO different cotangents; per parameter group: the replay's relative error against
the eager gradient, and whether the replayed gradient follows the cotangent at all.
usage: graph_probe.py [variant ...]     variants: default | nodb | deterministic | nofold | nofused | biassum | small"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(variant):
    if variant == "nodb":
        os.environ["TTDG_MIOPEN_DB"] = "0"
    import torch
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.modeling import backbone as bb
    from ttdg_mgm_amd.modeling import graphed
    if variant == "deterministic":
        torch.backends.cudnn.deterministic = True
    if variant == "nofold":
        bb.MULTI_FOLD = False
    if variant == "nofused":
        bb.FUSED_EPILOGUE = False
    if variant == "biassum":          # the bias gradient by a matrix-vector product instead of torch's multi-block reduction
        def backward(ctx, gout):
            gb = None
            if ctx.needs_input_grad[1]:
                g2 = gout.permute(0, 2, 3, 1).reshape(-1, gout.shape[1])
                gb = torch.ones(1, g2.shape[0], device=g2.device) @ g2
                gb = gb.reshape(-1)
            return gout, gb, (gout if ctx.has_res else None)
        ops.BiasAddFn.backward = staticmethod(backward)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = bb.FPN(freeze_at=2).to(dev).train()
    shape = (2, 3, 256, 256) if variant == "small" else (4, 3, 800, 800)
    x = torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last)
    params = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    with torch.no_grad():
        net(x)
    feats = ("p2", "p3", "p4", "p5", "p6")
    cots = []
    for seed in (1, 2):
        g = torch.Generator(device=dev).manual_seed(seed)
        with torch.no_grad():
            o = net(x)
        cots.append([torch.randn(o[k].shape, device=dev, generator=g).contiguous(memory_format=torch.channels_last) * 1e-3 for k in feats])

    def grads(fn, cot):
        for _, p in params:
            p.grad = None
        outs = fn(x)
        outs = [outs[k] for k in feats] if isinstance(outs, dict) else list(outs)
        torch.autograd.backward(outs, cot)
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() if p.grad is not None else None for n, p in params}

    eager = [grads(net, c) for c in cots]
    wrapped = graphed._TupleBackbone(net)
    wrapped.train()
    gfn = torch.cuda.make_graphed_callables(wrapped, (x.clone(),), num_warmup_iters=2)
    rep = [grads(gfn, c) for c in cots]
    rep.append(grads(gfn, cots[0]))          # the first cotangent again, after the second
    groups = {"res3": "bottom_up.res3", "res4": "bottom_up.res4", "res5": "bottom_up.res5", "fpn_weight": ("fpn_", ".weight"), "fpn_bias": ("fpn_", ".bias")}
    out = {"variant": variant, "shape": list(shape)}
    for gname, key in groups.items():
        names = [n for n, _ in params if (key in n if isinstance(key, str) else (key[0] in n and n.endswith(key[1])))]
        worst, follow, again, missing = 0.0, [], 0.0, 0
        for n in names:
            if eager[0][n] is None or rep[0][n] is None:
                missing += int((eager[0][n] is None) != (rep[0][n] is None))
                continue
            for k in (0, 1):
                den = float(eager[k][n].abs().max()) + 1e-30
                worst = max(worst, float((rep[k][n] - eager[k][n]).abs().max()) / den)
            de = float((eager[1][n] - eager[0][n]).abs().max()) + 1e-30
            follow.append(float((rep[1][n] - rep[0][n]).abs().max()) / de)
            again = max(again, float((rep[2][n] - rep[0][n]).abs().max()) / (float(eager[0][n].abs().max()) + 1e-30))
        out[gname] = {"tensors": len(names), "worst_rel_error_vs_eager": worst, "follows_cotangent_min": min(follow) if follow else None,
                      "follows_cotangent_max": max(follow) if follow else None, "same_cotangent_again_rel_diff": again, "grad_presence_mismatch": missing}
    print("PROBE " + json.dumps(out), flush=True)


if __name__ == "__main__":
    vs = sys.argv[1:] or ["default"]
    if len(vs) == 1:
        run(vs[0])
    else:
        import subprocess
        for v in vs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True)
            lines = [l for l in r.stdout.splitlines() if l.startswith("PROBE ")]
            print(lines[-1] if lines else "PROBE " + json.dumps({"variant": v, "failed": r.stderr[-600:]}), flush=True)
