import sys, copy
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import torch
from ttdg_mgm_amd import synth, ops
from ttdg_mgm_amd.modeling import backbone as bb
dev = torch.device("cuda:0")
g = synth.gen(7670)
torch.manual_seed(11)
net = bb.FPN(2).train()
for m in net.modules():
    if isinstance(m, bb.FrozenBatchNorm2d):
        m.weight.copy_(1.0 + 0.1 * torch.sin(torch.arange(m.weight.numel()).float()))
        m.running_mean.copy_(0.1 * torch.cos(torch.arange(m.weight.numel()).float()))
    if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
        m.bias.data.copy_(synth.normal(g, m.bias.shape, 0.1))
x0 = synth.normal(g, (2, 3, 96, 128), 1.0)
net64 = copy.deepcopy(net).double()
outs64 = net64(x0.double())
sum(v.square().mean() for v in outs64.values()).backward()
g64 = {n: p.grad for n, p in net64.named_parameters() if p.grad is not None}
bb.POINTWISE_MIN_PIXELS = 0
torch.backends.cudnn.deterministic = True
bb.OWN_POINTWISE = False
for fe, mf, cl in ((True, True, True),):
    bb.FUSED_EPILOGUE, bb.MULTI_FOLD, bb.CHANNELS_LAST = fe, mf, cl
    for trial in range(10):
        bb.OWN_POINTWISE = trial % 2 == 1
        nd = copy.deepcopy(net).to(dev)
        outs = nd(x0.to(dev))
        sum(v.square().mean() for v in outs.values()).backward()
        torch.cuda.synchronize()
        errs = {}
        for n, p in nd.named_parameters():
            if p.grad is None: continue
            t = g64[n]
            errs[n] = float((p.grad.cpu().double() - t).norm() / t.norm())
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print("own: fused_epilogue", fe, "multi_fold", mf, "channels_last", cl, "trial", trial, "n>1e-5:", sum(e > 1e-5 for e in errs.values()), [(n.replace("bottom_up.", ""), "%.1e" % e) for n, e in worst], flush=True)
