"""Diagnostics (not product): the GA-MGM solver alone on solver inputs saved by tools/gagm_trained_probe.py (trained-regime
A / Wds / U0).  Times the launch with HIP events (median of `reps`), prints iterations per stage and the phase shares, and
checks that the returned permutation equals the one the probe recorded.   usage: bench_gagm_inputs.py [inputs.pt] [reps] [debug flags] [threads]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ttdg_mgm_amd import ops  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "fixtures", "trained_solver_inputs.pt")
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    if len(sys.argv) > 3:
        from ttdg_mgm_amd import _lib
        ops.GAGM_VARIANT |= _lib.GAGM_LDS_PROJECTORS if int(sys.argv[3]) & 1 else 0
    if len(sys.argv) > 4:
        from ttdg_mgm_amd import _lib
        ops.GAGM_VARIANT |= _lib.GAGM_256_THREADS if int(sys.argv[4]) == 256 else 0
    dump = torch.load(path, weights_only=True)
    tot_us, tot_it = 0.0, 0
    for k, d in enumerate(dump):
        sizes = d["sizes"]
        apack, W, U0 = d["apack"].to(dev), d["Wds"].to(dev).contiguous(), d["U0"].to(dev).contiguous()
        gr = ops.graphs(sizes)
        times = []
        for prof in (False, True):
            cfg = ops.gagm_cfg(tau0=0.1, gamma=0.5, min_tau=1e-2, tol=1e-3, quad_weight=0.5, max_iter=200, sk_iter=20, profile=prof)
            for r in range(reps if not prof else 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                U, info, _ = ops.gagm_solve(apack, W, U0, gr, sizes, cfg)
                e1.record()
                torch.cuda.synchronize()
                if not prof:
                    times.append(e0.elapsed_time(e1) * 1e3)
        info = info.cpu().tolist()
        same = bool((U.cpu() == d["Ub"]).all())
        times.sort()
        med = times[len(times) // 2]
        ph = info[9:14]
        s = float(sum(ph)) or 1.0
        print("step %d sizes %s stages %s total %d  %.0f us  %.1f us/iter  same-as-probe %s  share B %.0f%% S %.0f%% V %.0f%% proj %.0f%% conv %.0f%%"
              % ((k, sizes, info[:6], info[6], med, med / max(info[6], 1), same) + tuple(100.0 * x / s for x in ph)), flush=True)
        tot_us += med
        tot_it += info[6]
    print("mean %.0f us per solve, %.1f us per iteration over %d solves" % (tot_us / len(dump), tot_us / max(tot_it, 1), len(dump)))


if __name__ == "__main__":
    main()
