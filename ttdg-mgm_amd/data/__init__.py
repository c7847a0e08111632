"""Test datasets (synthetic, or COCO-json via data/coco.py) + test loader (reference data/build.py:122-154:
InferenceSampler shard per rank, BatchSampler(TEST.BATCH, drop_last=False), trivial collate)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import synth

_REGISTRY = {}


def register_synthetic(name, num_images, size=512, cfg_id=2, kind="fundus", num_cls=2, id_offset=0):
    """``id_offset`` shifts the image ids (two streams evaluated by one DiceEvaluator must not share ids)."""
    _REGISTRY[name] = dict(n=num_images, size=size, cfg_id=cfg_id, kind=kind, num_cls=num_cls, id_offset=id_offset)


def register_coco_instances(name, metadata, json_file, image_root, input_format="RGB"):
    """Same call as detectron2's ``register_coco_instances`` (reference data/datasets/builtin.py:9-10); ``metadata`` is
    accepted for signature compatibility.  The json is parsed at registration, images are decoded when a shard is built."""
    from . import coco
    _REGISTRY[name] = dict(kind="coco", records=coco.load_coco_json(json_file, image_root), fmt=input_format,
                           json_file=json_file, image_root=image_root, metadata=dict(metadata or {}))
    _REGISTRY[name]["n"] = len(_REGISTRY[name]["records"])


def register_disk(name, root, source=None, workers=4):
    """A dataset of decoded images on local storage (data/disk.py): ``source`` (a registered dataset) is pre-rendered into
    ``root`` when given.  Streaming loaders read it with ``workers`` worker processes (cfg.DATALOADER.NUM_WORKERS in the
    reference, data/build.py:148-153)."""
    from . import disk
    n = disk.prerender(source, root) if source is not None else len([f for f in os.listdir(root) if f.endswith(".npz")])
    _REGISTRY[name] = dict(kind="disk", root=root, n=n, workers=int(workers))


def dataset_size(name):
    if name not in _REGISTRY:
        raise KeyError("Dataset '{}' is not registered! Available datasets are: {}".format(name, ", ".join(sorted(_REGISTRY))))
    return _REGISTRY[name]["n"]


def dataset_dicts(name, start=0, stop=None):
    """List of dicts: image (3,H,W) uint8, height, width, image_id, annotations [{bbox xyxy, category_id, mask bool}].
    ``start``/``stop`` generate only that index range (a rank's shard): images are synthesised on demand."""
    spec = _REGISTRY[name]
    if spec["kind"] == "disk":
        from . import disk
        ds = disk.DiskDataset(spec["root"], spec["n"])
        return [ds[i] for i in range(start, spec["n"] if stop is None else min(stop, spec["n"]))]
    if spec["kind"] == "coco":
        from . import coco
        return [coco.materialise(r, spec["fmt"]) for r in spec["records"][start:spec["n"] if stop is None else min(stop, spec["n"])]]
    out = []
    for i in range(start, spec["n"] if stop is None else min(stop, spec["n"])):
        seed = 1000 * spec["cfg_id"] + i
        gen = synth.fundus_image if spec["kind"] == "fundus" else synth.polyp_image
        img, boxes, classes, masks = gen(seed, spec["size"]) if spec["kind"] == "fundus" else gen(seed, spec["size"], spec["num_cls"])
        anns = [dict(bbox=boxes[k], category_id=int(classes[k]), mask=masks[k]) for k in range(len(classes))]
        out.append(dict(image=img, height=int(img.shape[1]), width=int(img.shape[2]), image_id=i + spec.get("id_offset", 0), annotations=anns,
                        seed=seed))
    return out


def mapped_size(h, w, min_size=800, max_size=1333):
    """ResizeShortestEdge [3P]: the shorter edge to ``min_size`` unless the longer one would pass ``max_size``."""
    s = min(min_size / min(h, w), max_size / max(h, w))
    return int(round(h * s)), int(round(w * s))


def map_for_test(d, min_size=800, max_size=1333, resize=True):
    """DatasetMapper(is_train=False) equivalent: resize the shorter edge to ``min_size`` (bilinear) and carry the
    teacher-forced detections (GT boxes jittered +-2 px, in resized coordinates).  ``resize=False`` leaves the pixels alone
    (``image`` is then the raw uint8 image, ``resize_to`` the target size): the streaming loader resizes on the device."""
    h, w = d["height"], d["width"]
    nh, nw = mapped_size(h, w, min_size, max_size)
    boxes = torch.stack([a["bbox"] for a in d["annotations"]]) if d["annotations"] else torch.zeros(0, 4)
    # teacher-forced detections: jittered ground-truth boxes for the seeded synthetic images, the plain ground truth otherwise
    tf = (synth.jitter_boxes(d["seed"] + 500000, boxes) if "seed" in d else boxes) * torch.tensor([nw / w, nh / h, nw / w, nh / h])
    out = dict(height=h, width=w, image_id=d["image_id"], tf_boxes=tf,
               tf_classes=torch.tensor([a["category_id"] for a in d["annotations"]], dtype=torch.int64), dataset_dict=d)
    if not resize:
        out["image"], out["resize_to"] = d["image"], (nh, nw)
        return out
    # bilinear; antialiased when shrinking, as the PIL resize of detectron2's ResizeShortestEdge is [3P] (no effect when enlarging)
    img = F.interpolate(d["image"][None].float(), size=(nh, nw), mode="bilinear", align_corners=False, antialias=nh < h or nw < w)[0]
    out["image"] = img.round().clamp(0, 255).to(torch.uint8)
    return out


STAGE_IN_THREAD = False      # disk streams: stage on a helper thread instead of the consumer's own (A/B switch, measured in DESIGN.md section 8)


class TestLoader:
    """Iterable over lists of mapped dicts; rank r sees the contiguous shard detectron2's InferenceSampler gives it.

    Two modes.  ``resident=True`` (bench.py, the parity tests): the shard is decoded, mapped and - with ``device`` -
    uploaded once, and every pass iterates over the resident items (the timed region starts with its inputs in HBM).
    ``resident=False`` (train_net.py, large COCO-json datasets): nothing is held; every pass decodes / synthesises,
    resizes and uploads batch by batch on a background thread, ``prefetch`` batches ahead, from pinned memory on a side
    stream - the loader the reference iterates twice per dataset (data/build.py:122-154, trainer.py:470,485).  Each item
    carries its ``dataset_dict`` (ground truth), so the evaluator never needs the whole dataset in memory."""

    def __init__(self, name, batch, rank=0, world=1, device=None, min_size=800, max_size=1333, resident=True, prefetch=2,
                 device_resize=None):
        n = dataset_size(name)
        shard = (n - 1) // world + 1 if n else 0           # detectron2 InferenceSampler [3P]: contiguous, unpadded
        self.name, self.batch, self.device = name, batch, device
        self.start, self.stop = min(shard * rank, n), min(shard * (rank + 1), n)
        self.min_size, self.max_size, self.resident, self.prefetch = min_size, max_size, resident, max(1, int(prefetch))
        self._dicts = self.items = None
        cuda = device is not None and torch.device(device).type == "cuda"
        # streaming to a GPU: upload the RAW uint8 image and resize there (csrc/resize.hip) unless told otherwise
        self.device_resize = (cuda and not resident) if device_resize is None else bool(device_resize and cuda)
        self._disk = None
        # disk streams: stage (DMA from the ring + resize launch) from the consumer's own thread one batch ahead (default), or on a helper
        # thread (data.STAGE_IN_THREAD = True).  Measured back to back on the final build: 0.932 / 0.950 x the resident rate against 0.902 / 0.940
        self.stage_in_thread = STAGE_IN_THREAD
        if not resident and _REGISTRY[name]["kind"] == "disk":
            from . import disk
            spec = _REGISTRY[name]
            self._disk = disk.DiskStream(spec["root"], spec["n"], batch, workers=spec["workers"], prefetch=self.prefetch, ring=cuda)
        if resident:
            self._dicts = dataset_dicts(name, self.start, self.stop)
            self.items = [map_for_test(d, min_size, max_size) for d in self._dicts]
            if device is not None:                                   # keep inputs resident in HBM (bench)
                for it in self.items:
                    it["image"] = it["image"].to(device)

    @property
    def dataset_dicts(self):
        """Ground truth of the shard (materialised on first use in the streaming mode; the evaluator does not need it
        there: it reads the ``dataset_dict`` every item carries)."""
        if self._dicts is None:
            self._dicts = dataset_dicts(self.name, self.start, self.stop)
        return self._dicts

    def __len__(self):
        return (self.stop - self.start + self.batch - 1) // self.batch

    def _upload(self, t, tag):
        """Host tensor -> device through PERSISTENT pinned staging buffers (three per tag, used round-robin; a buffer is
        rewritten only after the copy that read it has completed).  ``tensor.pin_memory()`` per batch measured 7.5 ms per call on
        the GPU box - a fresh pinned allocation each time - against 0.3 ms for the memcpy into a kept buffer."""
        ring = self.__dict__.setdefault("_pin", {}).setdefault(tag, dict(bufs=[None] * 3, evs=[None] * 3, k=0))
        k = ring["k"]
        ring["k"] = (k + 1) % 3
        n = t.numel() * t.element_size()
        if ring["evs"][k] is not None:
            ring["evs"][k].synchronize()
        if ring["bufs"][k] is None or ring["bufs"][k].numel() < n:
            ring["bufs"][k] = torch.empty(max(n, 1 << 20), dtype=torch.uint8).pin_memory()
        host = ring["bufs"][k][:n].view(t.dtype).view(t.shape)
        host.copy_(t)
        dev = host.to(self.device, non_blocking=True)
        ring["evs"][k] = torch.cuda.Event()
        ring["evs"][k].record()
        return dev

    def start_workers(self):
        """Spawn the worker processes of a disk stream (loader construction: outside any timed region)."""
        if self._disk is not None:
            self._disk.start()

    def _load_batch(self, lo, hi, stream, dicts=None):
        stacked = masks = None
        slot = None
        if isinstance(dicts, dict):                       # a batch collated by a disk-stream worker
            from . import disk
            slot = dicts.get("slot")
            items_ = disk.expand(dicts, self._disk.ring)
            stacked, masks, dicts = dicts["images"], dicts["masks"], items_
        direct = slot is not None and self._disk.ring is not None and self._disk.ring.pinned      # the slot is page-locked: DMA straight from it
        if dicts is None:
            dicts = dataset_dicts(self.name, lo, hi)
        items = [map_for_test(d, self.min_size, self.max_size, resize=not self.device_resize) for d in dicts]
        ev = None
        if self.device is not None and torch.device(self.device).type == "cuda":
            with torch.cuda.stream(stream):
                if self.device_resize:
                    from .. import ops
                    same = len({(tuple(it["image"].shape), it["resize_to"]) for it in items}) == 1
                    if same:      # one pinned upload + one resize launch for the batch
                        raw = stacked.to(self.device, non_blocking=True) if direct else \
                            self._upload(stacked if stacked is not None else torch.stack([it["image"] for it in items]), "img")
                        out = ops.resize_u8(raw, *items[0]["resize_to"])
                        for k, it in enumerate(items):
                            it["image"] = out[k]
                    else:
                        for it in items:
                            it["image"] = ops.resize_u8(it["image"].contiguous().pin_memory().to(self.device, non_blocking=True), *it["resize_to"])
                else:
                    for it in items:
                        it["image"] = it["image"].pin_memory().to(self.device, non_blocking=True)
                # the evaluator's inputs travel the same way: ground-truth masks pinned and uploaded on the side stream, so that
                # the Dice pass never issues a pageable (= synchronous) copy on the compute stream
                if masks is not None:                    # one upload for the ground truth of the whole batch
                    dm, k = (masks.to(self.device, non_blocking=True) if direct else self._upload(masks, "gt")), 0
                    for it in items:
                        n = len(it["dataset_dict"]["annotations"])
                        if n:
                            it["dataset_dict"]["device_masks"] = dm[k:k + n]
                            if direct:       # the host masks are VIEWS of the ring slot, which goes back to a worker once this upload
                                # has completed: nobody may read them later (evaluator fallbacks) - hand out the device copy instead
                                for j, a in enumerate(it["dataset_dict"]["annotations"]):
                                    a["mask"] = dm[k + j]
                        k += n
                else:
                    for it in items:
                        anns = it["dataset_dict"]["annotations"]
                        if anns and "device_masks" not in it["dataset_dict"]:
                            it["dataset_dict"]["device_masks"] = torch.stack([a["mask"] for a in anns]).pin_memory().to(self.device, non_blocking=True)
                if direct:
                    for it in items:
                        it["dataset_dict"].pop("image", None)      # a view of the ring slot as well; the batch carries the device image
                ev = torch.cuda.Event()
                ev.record(stream)
                if slot is not None and self._disk.ring is not None:
                    self._disk.ring.events[slot] = ev          # the slot may be rewritten once its uploads have completed
        return items, ev

    def __iter__(self):
        if self.resident:
            for i in range(0, len(self.items), self.batch):
                yield self.items[i:i + self.batch]
            return
        cuda = self.device is not None and torch.device(self.device).type == "cuda"
        stream = torch.cuda.Stream(device=self.device) if cuda else None
        if self._disk is not None and not self.stage_in_thread:
            # Decoding happens in the worker PROCESSES; what is left for this process - pin, upload, resize launch - is a
            # handful of calls, issued from the consumer's own thread one batch ahead (the copies and the resize run on the side
            # stream under the previous step's kernels).  No producer thread: a second Python thread has to win the GIL from a
            # main thread that is launching ~1500 kernels per step, and measured 31 instead of 84 images/s that way.
            it = self._disk.epoch(self.start, self.stop)

            def staged():
                try:
                    return self._load_batch(0, 0, stream, next(it))
                except StopIteration:
                    return None
            nxt = staged()
            while nxt is not None:
                items, ev = nxt
                nxt = staged()                     # batch k + 1 goes to the side stream before batch k is consumed
                if ev is not None:
                    cur = torch.cuda.current_stream()
                    cur.wait_event(ev)
                    for x in items:
                        x["image"].record_stream(cur)
                        dm = x["dataset_dict"].get("device_masks")
                        if dm is not None:
                            dm.record_stream(cur)
                yield items
            return
        import queue
        import threading
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def put(x):
            """Blocking put that gives up when the consumer has gone away (early `break`, exception, `limit`)."""
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                if cuda:
                    torch.cuda.set_device(self.device)
                if self._disk is not None:          # decoded by the worker processes; this thread pins, uploads, resizes
                    for dicts in self._disk.epoch(self.start, self.stop):
                        if stop.is_set() or not put(self._load_batch(0, 0, stream, dicts)):
                            return
                else:
                    for lo in range(self.start, self.stop, self.batch):
                        if stop.is_set() or not put(self._load_batch(lo, min(lo + self.batch, self.stop), stream)):
                            return
                put(None)
            except BaseException as e:          # surfaced on the consumer's thread
                put(e)

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                got = q.get()
                if got is None:
                    break
                if isinstance(got, BaseException):
                    raise got
                items, ev = got
                if ev is not None:
                    cur = torch.cuda.current_stream()
                    cur.wait_event(ev)                           # the upload ran on the side stream
                    for it in items:                             # allocated on the side stream, consumed on this one: the
                        it["image"].record_stream(cur)           # caching allocator must not recycle it under queued kernels
                        dm = it["dataset_dict"].get("device_masks")
                        if dm is not None:
                            dm.record_stream(cur)
                yield items
                del items, got
        finally:
            # consumer finished or left early (GeneratorExit / exception): release the producer and whatever it queued
            stop.set()
            while True:
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
            t.join(timeout=30)


def build_detection_test_loader(cfg, dataset_name, rank=0, world=1, device=None, resident=True):
    return TestLoader(dataset_name, cfg.TEST.BATCH, rank, world, device,
                      cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, resident=resident)
