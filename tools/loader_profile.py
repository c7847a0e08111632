"""Diagnostics: cProfile of the streaming loader loop on the GPU box, with a model-sized amount of main-thread work absent."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ttdg_mgm_amd import data  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    n, B = 80, 4
    data.register_synthetic("lp_src", n, size=512, cfg_id=22)
    data.register_disk("lp", os.path.join(tempfile.gettempdir(), "lp_stream"), source="lp_src", workers=4)
    ld = data.TestLoader("lp", B, 0, 1, dev, 800, 1333, resident=False)
    ld.start_workers()
    for b in ld:
        pass
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    k = 0
    for b in ld:
        k += len(b)
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    print("%d images in %.3f s = %.1f images/s" % (k, dt, k / dt))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    # per-call wall times inside _load_batch
    it = ld._disk.epoch(0, n)
    s = torch.cuda.Stream()
    tt = []
    t0 = time.perf_counter()
    for dicts in it:
        t1 = time.perf_counter()
        items, ev = ld._load_batch(0, 0, s, dicts)
        t2 = time.perf_counter()
        tt.append((t1 - t0, t2 - t1))
        t0 = time.perf_counter()
    print("per batch: wait for the worker %.2f ms (median), stage %.2f ms (median)" % (sorted(x[0] for x in tt)[len(tt) // 2] * 1e3, sorted(x[1] for x in tt)[len(tt) // 2] * 1e3))
    print("first 6:", [(round(a * 1e3, 1), round(b * 1e3, 1)) for a, b in tt[:6]])


if __name__ == "__main__":
    main()
