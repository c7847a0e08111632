"""Diagnostics: where does the streaming loader spend its time on the GPU box?  (disk stream alone / + pin + H2D + device resize /
through TestLoader), with a live HIP context in the parent as in bench.py."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ttdg_mgm_amd import data, ops  # noqa: E402
from ttdg_mgm_amd.data import disk  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    n, B = 80, 4
    data.register_synthetic("lp_src", n, size=512, cfg_id=22)
    root = os.path.join(tempfile.gettempdir(), "lp_stream")
    t0 = time.perf_counter()
    data.register_disk("lp", root, source="lp_src", workers=4)
    print("prerender %.2f s" % (time.perf_counter() - t0))
    st = disk.DiskStream(root, n, B, workers=4)
    t0 = time.perf_counter()
    st.start()
    print("worker start %.2f s" % (time.perf_counter() - t0))
    for rep in range(2):
        t0 = time.perf_counter()
        k = sum(len(b["meta"]) for b in st.epoch(0, n))
        print("disk epoch: %d images, %.1f images/s" % (k, k / (time.perf_counter() - t0)))
    ld = data.TestLoader("lp", B, 0, 1, dev, 800, 1333, resident=False)
    ld.start_workers()
    for rep in range(2):
        t0 = time.perf_counter()
        k = 0
        for b in ld:
            k += len(b)
        torch.cuda.synchronize()
        print("TestLoader (pin + H2D + device resize): %d images, %.1f images/s" % (k, k / (time.perf_counter() - t0)))
    # stages of one batch
    dicts = disk.expand([b for b in st.epoch(0, B)][0])
    s = torch.cuda.Stream()
    for rep in range(3):
        t0 = time.perf_counter()
        items = [data.map_for_test(d, 800, 1333, resize=False) for d in dicts]
        t1 = time.perf_counter()
        raw = torch.stack([it["image"] for it in items])
        t2 = time.perf_counter()
        pinned = raw.pin_memory()
        t3 = time.perf_counter()
        with torch.cuda.stream(s):
            up = pinned.to(dev, non_blocking=True)
            out = ops.resize_u8(up, 800, 800)
        s.synchronize()
        t4 = time.perf_counter()
        print("one batch: map %.2f ms, stack %.2f ms, pin %.2f ms, H2D + resize %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))


if __name__ == "__main__":
    main()
