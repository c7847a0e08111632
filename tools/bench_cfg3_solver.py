"""cfg-3 (8 graphs x 256 nodes) GA-MGM solve in pieces: where do the microseconds of an iteration go?  The solver's own inputs
(A, Wds, U0) are taken from one MGM3_unsup forward on the cfg-3 synthetic nodes; then ttdg_gagm_solve is timed (HIP events,
median of `reps`) for the whole schedule, for the Sinkhorn stages alone (cfg.max_stages = 5) and for a Hungarian stage entered
directly (cfg.start_hungarian) -> microseconds per Sinkhorn-stage iteration and per Hungarian-stage iteration.
    python tools/bench_cfg3_solver.py [sizes like 256x256x...] [reps]"""
import json
import statistics
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402
from ttdg_mgm_amd import _lib, ops, synth  # noqa: E402
from ttdg_mgm_amd.GModule import MGM3_unsup  # noqa: E402

dev = torch.device("cuda:0")
sizes = tuple(int(x) for x in sys.argv[1].split("x")) if len(sys.argv) > 1 else (256,) * 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ops.GAGM_VARIANT = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # e.g. 16 = round 3's column-per-thread projector
nodes, labels = synth.node_sets(5, sizes, scale=0.5)
params, U = synth.mgm3_params(6), synth.universe(7)
m = MGM3_unsup(2, 32).to(dev).eval()
m.load_state_dict(params)
tr = {}
with torch.no_grad():
    m([x.to(dev) for x in nodes], [l.to(dev) for l in labels], U.to(dev), trace=tr)
apack, W, U0 = tr["apack"], tr["Wds"], tr["U0"]
gr = ops.graphs(list(sizes))
M = sum(sizes)


def timed(cfg):
    ts, info = [], None
    for r in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        Ub, info, _ = ops.gagm_solve(apack, W, U0, gr, list(sizes), cfg)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    it = info.cpu().tolist()
    return statistics.median(ts), it


out = {"sizes": sizes, "M": M, "variant": ops.GAGM_VARIANT}
t_all, it_all = timed(ops.gagm_cfg())
t_sk, it_sk = timed(ops.gagm_cfg(max_stages=5))
t_h, it_h = timed(ops.gagm_cfg(start_hungarian=True))
t_all1, it_all1 = timed(ops.gagm_cfg(variant=_lib.GAGM_ONE_LAUNCH))
t_sk1, _ = timed(ops.gagm_cfg(max_stages=5, variant=_lib.GAGM_ONE_LAUNCH))
n_sk, n_h = sum(it_sk[:5]), it_h[0]
per_iter_bytes = 4.0 * (sum(n * n for n in sizes) + M * M + 3 * M * 32)          # SURVEY.md §8d, A6 per iteration
out["full"] = dict(us=t_all, iters=it_all[:6], total=it_all[6], us_per_iter=t_all / max(1, it_all[6]),
                   algorithmic_GBps=per_iter_bytes * it_all[6] / (t_all * 1e-6) / 1e9, frac_hbm=per_iter_bytes * it_all[6] / (t_all * 1e-6) / 8e12)
out["sinkhorn_stages"] = dict(us=t_sk, iters=it_sk[:5], us_per_iter=t_sk / max(1, n_sk))
out["one_launch"] = dict(full_us=t_all1, iters=it_all1[:6], sinkhorn_stages_us=t_sk1)
out["hungarian_stage_from_U0"] = dict(us=t_h, iters=n_h, us_per_iter=t_h / max(1, n_h))
# every iteration a LAP (cycle shortcut off), duals carried from one iteration to the next; the same with the one-wavefront scipy-order LAP
for name, var in (("certified", 0), ("scipy_order", _lib.GAGM_SCIPY_ORDER_LAP), ("certified_one_launch", _lib.GAGM_ONE_LAUNCH)):
    t, it = timed(ops.gagm_cfg(start_hungarian=True, max_stages=1, no_cycle_skip=True, max_iter=24, variant=var))
    out["hungarian_24_iterations_" + name] = dict(us=t, iters=it[0], us_per_iter=t / max(1, it[0]), certified=it[12], fallbacks=it[13])
# where the cycles of the projection launch go (in-kernel cycle counters, summed over the 8 graphs and all iterations; units of 1024 cycles)
for name, kw in (("full", {}), ("hungarian_24", dict(start_hungarian=True, max_stages=1, no_cycle_skip=True, max_iter=24))):
    for mode in (1, 2):
        _, it = timed(ops.gagm_cfg(profile=mode, **kw))
        out.setdefault("project_kernel_kcycles_" + name, {})["operands+S, V, projector, norms" if mode == 1 else "Sinkhorn, certified LAP, scipy-order LAP, norms"] = it[16:20] + [dict(certified=it[12], fallbacks=it[13], iterations=it[6])]
    _, it = timed(ops.gagm_cfg(profile=3, **kw))
    out["project_kernel_kcycles_" + name]["pricing rounds, rows to augment, Dijkstra steps, max certified-LAP kcycles, max scipy-order kcycles"] = it[16:20] + [it[20]]
print(json.dumps(out, indent=1))
