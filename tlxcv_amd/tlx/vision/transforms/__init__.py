"""`tensorlayerx.vision.transforms` subset: Compose, Resize, Normalize, ToTensor (+ CentralCrop, HWC2CHW).

Host-side image preprocessing, exactly where the reference does it (PIL/numpy on the CPU, one image at a
time, demo/image_classification/predict.py:21-29) — it is not on the accelerated path.  ToTensor hands the
result to the device the engine runs on.  Semantics restated from the TensorLayerX documentation
[TLX-recalled]: images are HWC arrays; Resize is bilinear by default; Normalize is (x - mean) / std per
channel; ToTensor rescales by 1/255 only for uint8 input and reorders to CHW when asked.
"""
import numpy as np
import torch

from . import utils  # noqa: F401

__all__ = ["Compose", "Resize", "Normalize", "ToTensor", "CentralCrop", "HWC2CHW"]


class Compose:
    """Compose([Resize, Normalize, ToTensor]) — the pipeline of every inference demo (predict.py:22-28) — runs as ONE device
    launch when a GPU is present (uint8 image up, tlxmi_preprocess_u8: Pillow's resampler restated in integer arithmetic,
    so the tensor is bit-identical to what the host transforms below produce); any other composition, or no GPU, runs
    the host transforms one by one.  `batch(images)` does a whole (N, H, W, C) uint8 batch in one launch."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def _device_plan(self):
        ts = self.transforms
        if len(ts) == 3 and isinstance(ts[0], Resize) and isinstance(ts[1], Normalize) and isinstance(ts[2], ToTensor):
            return ts[0], ts[1], ts[2]
        if len(ts) == 2 and isinstance(ts[0], Resize) and isinstance(ts[1], ToTensor):
            return ts[0], None, ts[1]
        return None

    def batch(self, images, dtype=torch.float32, fold=0):
        """(N, H, W, C) uint8 (numpy, or a torch tensor on host or device) -> the batch tensor, on the device."""
        from .... import engine as E
        plan = self._device_plan()
        if plan is None:
            raise NotImplementedError("Compose.batch: only [Resize, Normalize, ToTensor] / [Resize, ToTensor] run on the device")
        rs, nm, tt = plan
        if isinstance(images, torch.Tensor):
            t = images
        else:
            arr = np.ascontiguousarray(images)
            t = torch.from_numpy(arr if arr.flags.writeable else arr.copy())
        if t.dim() == 3:
            t = t.unsqueeze(0)
        return E.preprocess_u8(t.cuda(), rs.size, None if nm is None else nm.mean, None if nm is None else nm.std,
                               layout=tt.data_format, dtype=dtype, interpolation=rs.interpolation, fold=fold)

    def __call__(self, data):
        # the device path takes 1- and 3-channel uint8 images only: on RGBA Pillow premultiplies alpha around the resize,
        # which tlxmi_preprocess_u8 does not restate — those (and every other input) run the host transforms below
        a = data if isinstance(data, np.ndarray) else None
        if (a is not None and a.dtype == np.uint8 and a.ndim == 3 and a.shape[-1] in (1, 3) and torch.cuda.is_available()
                and self._device_plan() is not None and self.transforms[0].interpolation in ("bilinear", "bicubic")):
            return self.batch(a)[0]
        for t in self.transforms:
            data = t(data)
        return data


def _as_hwc(img):
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    return a


class Resize:
    def __init__(self, size, interpolation="bilinear"):
        self.size = (size, size) if isinstance(size, int) else tuple(size)    # (h, w)
        self.interpolation = interpolation

    def __call__(self, img):
        from PIL import Image
        a = _as_hwc(img)
        mode = {"bilinear": Image.BILINEAR, "nearest": Image.NEAREST, "bicubic": Image.BICUBIC}[self.interpolation]
        h, w = self.size
        if a.dtype == np.uint8:
            pil = Image.fromarray(a.squeeze(-1) if a.shape[-1] == 1 else a)
            out = np.asarray(pil.resize((w, h), mode))
            return out[:, :, None] if out.ndim == 2 else out
        chans = [np.asarray(Image.fromarray(a[:, :, c].astype(np.float32), mode="F").resize((w, h), mode))
                 for c in range(a.shape[-1])]
        return np.stack(chans, -1)


class CentralCrop:
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        a = _as_hwc(img)
        h, w = self.size
        top, left = (a.shape[0] - h) // 2, (a.shape[1] - w) // 2
        return a[top:top + h, left:left + w]


class Normalize:
    def __init__(self, mean, std):
        self.mean = np.asarray(mean, dtype=np.float32)
        self.std = np.asarray(std, dtype=np.float32)

    def __call__(self, img):
        a = _as_hwc(img).astype(np.float32)
        return (a - self.mean) / self.std


class HWC2CHW:
    def __call__(self, img):
        return np.transpose(_as_hwc(img), (2, 0, 1))


class ToTensor:
    def __init__(self, data_format="HWC"):
        if data_format not in ("HWC", "CHW"):
            raise ValueError("data_format should be CHW or HWC. Got {}".format(data_format))
        self.data_format = data_format

    def __call__(self, img):
        a = _as_hwc(img)
        if a.dtype == np.uint8:
            a = a.astype(np.float32) / 255.0
        a = a.astype(np.float32)
        if self.data_format == "CHW":
            a = np.transpose(a, (2, 0, 1))
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.cuda() if torch.cuda.is_available() else t
