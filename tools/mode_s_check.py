"""Mode S validation on ONE GPU: launch with
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/mode_s_check.py
Two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device).  Checks that one synchronous-universe
adaptation step on 2 ranks x 2 images IS the single-process step on the same 4 images (loss, every updated parameter),
that the replicas end up bit-identical, and that a rank without a batch (inputs=None) changes nothing.  The matching
solver is rounding-chaotic at random weights, so the single-process run's pseudo-labels are fed to the
Mode S run on every rank (forced_U, broadcast from rank 0): everything else is smooth in the inputs.  Prints one JSON line on rank 0."""
import copy
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def force(model, Ub):
    orig = model.multi_matching_unsup.forward
    model.multi_matching_unsup.forward = lambda nodes, labels, U, _o=orig: _o(nodes, labels, U, forced_U=Ub)


def main():
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    if os.environ.get("MODE_S_DETERMINISTIC", "1") != "0":
        # the vendor's convolution backward in its deterministic mode on BOTH sides (MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC
        # through torch.backends.cudnn.deterministic): the comparison is about Mode S, not about MIOpen's atomics
        torch.backends.cudnn.deterministic = True
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.engine import BaselineTrainer
    from ttdg_mgm_amd.modeling import calibrate_frozen_bn
    cfg = get_cfg()
    cfg.TEST.BATCH = 4
    cfg.INPUT.MIN_SIZE_TEST = 384
    cfg.MODEL.DEVICE = "cuda:0"
    data.register_synthetic("ms_ds", 4, size=256)
    torch.manual_seed(0)
    model = BaselineTrainer.build_model(cfg)
    model.teacher_forced = True
    model.train()
    full = next(iter(data.build_detection_test_loader(cfg, "ms_ds")))
    calibrate_frozen_bn(model, full)
    start = copy.deepcopy(model.state_dict())

    # single-process step on all four images
    ref = copy.deepcopy(model)
    tr = {}
    orig = ref.multi_matching_unsup.forward
    ref.multi_matching_unsup.forward = lambda nodes, labels, U, _o=orig: _o(nodes, labels, U, trace=tr)
    loss_ref = BaselineTrainer.tta_step(ref, BaselineTrainer.build_optimizer(cfg, ref), full)
    # the vendor convolutions are not bit-reproducible run to run, so each rank's free-running solve may settle on different
    # pseudo-labels: rank 0's are THE pseudo-labels, and rank 0's single-process result is the one Mode S is compared with
    Ub = tr["Ub"].cpu()
    dist.broadcast(Ub, 0)
    Ub = Ub.cuda()
    ref_params = {k: v.detach().clone() for k, v in ref.named_parameters()}
    ref_grads = {k: v.grad.detach().clone() for k, v in ref.named_parameters() if v.grad is not None}

    def ulp(x):                       # spacing of fp32 at |x| (a parameter update rounds to the parameter's own grid)
        a = x.abs()
        return torch.nextafter(a, torch.full_like(a, float("inf"))) - a

    def compare(m):
        """Every figure of a step against the single-process step: worst parameter / gradient difference, and the worst parameter
        difference BEYOND 4 ulp of the parameter itself (p - lr*buf rounds to p's grid: 1 ulp of an O(1) weight is 1.2e-7
        whatever the size of the update)."""
        moved, worst, wname, gworst, gname, gmax, beyond = 0.0, 0.0, "", 0.0, "", 0.0, 0.0
        same_set = True
        for k, p in m.named_parameters():
            dd = (p.detach() - ref_params[k]).abs()
            d = float(dd.max())
            beyond = max(beyond, float((dd - 4.0 * ulp(ref_params[k])).clamp_min(0).max()))
            if d > worst:
                worst, wname = d, k
            moved = max(moved, float((ref_params[k] - start[k]).abs().max()))
            same_set &= (p.grad is not None) == (k in ref_grads)
            if p.grad is not None and k in ref_grads:
                gd = float((p.grad - ref_grads[k]).abs().max())
                gmax = max(gmax, float(ref_grads[k].abs().max()))
                if gd > gworst:
                    gworst, gname = gd, k
        return {"max_param_diff_vs_single_process": worst, "max_param_diff_beyond_4ulp": beyond, "worst_param": wname, "max_param_update": moved,
                "max_grad_diff": gworst, "worst_grad": gname, "max_grad": gmax, "same_gradient_set": bool(same_set)}

    out = {}
    # the denominator: the SAME single-process step once more (same start, same pseudo-labels) - what the vendor's
    # convolution backward alone moves between two runs of one program
    again = copy.deepcopy(model)
    again.load_state_dict(start)
    force(again, Ub)
    BaselineTrainer.tta_step(again, BaselineTrainer.build_optimizer(cfg, again), full)
    torch.cuda.synchronize()
    out["single_process_rerun"] = compare(again)
    del again
    for name, local in (("split", full[2 * rank:2 * rank + 2]), ("idle_rank", full if rank == 0 else None)):
        m = copy.deepcopy(model)
        m.load_state_dict(start)
        m.sync_universe = True
        force(m, Ub)
        loss = BaselineTrainer.tta_step(m, BaselineTrainer.build_optimizer(cfg, m), local)
        torch.cuda.synchronize()
        digest = torch.tensor([sum(float(p.detach().double().sum()) for p in m.parameters()), float(loss.detach())], dtype=torch.float64)
        both = [torch.zeros_like(digest) for _ in range(2)]
        dist.all_gather(both, digest)
        out[name] = dict(compare(m), loss=float(loss.detach()), loss_single_process=float(loss_ref.detach()),
                         replicas_identical=bool(both[0][0] == both[1][0]), replicated_loss_identical=bool(both[0][1] == both[1][1]))
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
