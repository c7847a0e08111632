"""Minimal Boxes / Instances / ImageList with the attribute surface the path touches (detectron2 [3P] look-alikes:
``len(inst)``, ``inst._fields``, ``inst.pred_boxes.tensor``, ``inst.pred_classes`` — build_graph.py:79-85)."""
import torch


class Boxes:
    def __init__(self, tensor):
        self.tensor = tensor.reshape(-1, 4) if tensor.numel() == 0 else tensor

    def __len__(self):
        return self.tensor.shape[0]

    def clip(self, size):
        h, w = size
        t = self.tensor
        self.tensor = torch.stack((t[:, 0].clamp(0, w), t[:, 1].clamp(0, h), t[:, 2].clamp(0, w), t[:, 3].clamp(0, h)), dim=1)

    def nonempty(self, threshold=0.0):
        t = self.tensor
        return ((t[:, 2] - t[:, 0]) > threshold) & ((t[:, 3] - t[:, 1]) > threshold)

    def area(self):
        t = self.tensor
        return (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])

    def scale(self, sx, sy):
        self.tensor = self.tensor * self.tensor.new_tensor([sx, sy, sx, sy])


class Instances:
    def __init__(self, image_size, **fields):
        object.__setattr__(self, "_image_size", tuple(image_size))
        object.__setattr__(self, "_fields", {})
        for k, v in fields.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def __setattr__(self, name, value):
        if name.startswith("_"):
            object.__setattr__(self, name, value)
        else:
            self.set(name, value)

    def __getattr__(self, name):
        f = object.__getattribute__(self, "_fields")
        if name in f:
            return f[name]
        raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    def to(self, device):
        out = Instances(self._image_size)
        for k, v in self._fields.items():
            out.set(k, Boxes(v.tensor.to(device)) if isinstance(v, Boxes) else (v.to(device) if hasattr(v, "to") else v))
        return out

    def __getitem__(self, idx):
        out = Instances(self._image_size)
        for k, v in self._fields.items():
            out.set(k, Boxes(v.tensor[idx]) if isinstance(v, Boxes) else v[idx])
        return out


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @staticmethod
    def from_tensors(tensors, size_divisibility=0):
        hs, ws = [t.shape[-2] for t in tensors], [t.shape[-1] for t in tensors]
        H, W = max(hs), max(ws)
        if size_divisibility > 1:
            d = size_divisibility
            H, W = (H + d - 1) // d * d, (W + d - 1) // d * d
        out = tensors[0].new_zeros((len(tensors), tensors[0].shape[0], H, W))
        for i, t in enumerate(tensors):
            out[i, :, :t.shape[-2], :t.shape[-1]] = t
        return ImageList(out, [(h, w) for h, w in zip(hs, ws)])
