"""The detection demo's input pipeline — `Resize(size, max_size, auto_divide)`, `Normalize(mean, std)`, `ToTensor` over
`(image, label)` pairs (demo/object_detection/transforms.py:96-246, used at predict-YOLOv3.py:54-61 and the DETR / SSD demos).

The reference file lives in the demo directory and resizes with `cv2.resize(..., INTER_LINEAR)` on the host, one image at a
time.  Here the same classes compose into ONE device launch (tlxmi_preprocess_linear_u8): `Compose([Resize, Normalize])`
on a uint8 HWC image returns the normalised image already on the GPU.  A script switches by importing these names instead
of its local `transforms` module.  OpenCV's 8-bit bilinear resize is restated (two taps per axis, 11-bit fixed-point
weights, the coefficient loop of resize.cpp in float32) — UNPINNED: cv2 is not in this image, and an OpenCV built with
IPP / another HAL may differ in the last bit.  Label handling (boxes / area / size / im_shape / scale_factor / orig_size)
follows :163-199; segmentation masks (cv2 nearest-neighbour resize of the target masks, :190-199) are a training-time
target transform and raise NotImplementedError.
"""
import numpy as np
import torch

__all__ = ["Resize", "Normalize", "ToTensor", "Compose", "make_divided", "output_size", "linear_tables"]

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def make_divided(x, divided=8):                       # transforms.py:297-301
    if divided:
        d = x % divided
        x += divided - d if d else 0
    return x


def output_size(image_hw, size, max_size=None, auto_divide=None):
    """(out_h, out_w) of Resize._resize for an image of (h, w) (transforms.py:114-150).  The reference computes the pair in
    cv2's (width, height) order by swapping the roles of h and w; the arithmetic below is the same, un-swapped."""
    h, w = int(image_hw[0]), int(image_hw[1])
    if isinstance(size, (list, tuple)):
        ow, oh = int(size[0]), int(size[1])           # passed to cv2.resize as (width, height), :136-137
    else:
        shape = size
        if max_size is not None:
            lo, hi = float(min(w, h)), float(max(w, h))
            if hi / lo * shape > max_size:
                shape = int(round(max_size * lo / hi))
        if (h <= w and h == shape) or (w <= h and w == shape):
            oh, ow = h, w
        elif h < w:
            oh, ow = shape, int(shape * w / h)
        else:
            ow, oh = shape, int(shape * h / w)
    if auto_divide:
        oh, ow = make_divided(oh, auto_divide), make_divided(ow, auto_divide)
    return oh, ow


def linear_tables(n_in, n_out):
    """Tap indices [n_out][2] (clamped into the image) and fixed-point weights [n_out][2] of cv2.resize INTER_LINEAR along one
    axis: fx = float((d + 0.5) * scale - 0.5), s = floor(fx), weights round-half-even((1 - f, f) * 2048), f forced to 0
    where the window leaves the image."""
    scale = float(n_in) / float(n_out)
    d = np.arange(n_out, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int64)
    f = (fx - s.astype(np.float32)).astype(np.float32)
    low = s < 0
    f[low], s[low] = 0.0, 0
    high = s >= n_in - 1
    f[high], s[high] = 0.0, n_in - 1
    w1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int32)              # cvRound: round half to even
    w0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_SCALE)).astype(np.int32)
    idx = np.stack([s, np.minimum(s + 1, n_in - 1)], 1).astype(np.int32)
    return np.ascontiguousarray(idx), np.ascontiguousarray(np.stack([w0, w1], 1).astype(np.int32))


def _resize_host(image, oh, ow):
    """The same arithmetic on the host (numpy): the fallback when no GPU is present and what the device kernel must equal."""
    a = np.asarray(image)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    xi, xa = linear_tables(a.shape[1], ow)
    yi, yb = linear_tables(a.shape[0], oh)
    s = a.astype(np.int32)
    rows = s[:, xi[:, 0], :] * xa[None, :, 0, None] + s[:, xi[:, 1], :] * xa[None, :, 1, None]        # (H, ow, C), scale 2^11
    r0, r1 = rows[yi[:, 0]] >> 4, rows[yi[:, 1]] >> 4
    v = (((yb[:, 0, None, None] * r0) >> 16) + ((yb[:, 1, None, None] * r1) >> 16) + 2) >> 2
    out = np.clip(v, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


class Resize(object):
    """transforms.py:96-199."""

    def __init__(self, size, max_size, auto_divide=None):
        self.size, self.max_size, self.auto_divide = size, max_size, auto_divide

    def out_hw(self, image):
        return output_size(image.shape[:2], self.size, self.max_size, self.auto_divide)

    def target(self, image, out_hw, target):
        oh, ow = out_hw
        rh, rw = float(oh) / float(image.shape[0]), float(ow) / float(image.shape[1])
        target = target.copy() if target else {}
        if "orig_size" not in target:
            h, w = image.shape[:2]
            target["orig_size"] = np.asarray((w, h), dtype=np.int64)
        if "boxes" in target:
            target["boxes"] = target["boxes"] * np.asarray([rw, rh, rw, rh], dtype=np.float32)
        if "area" in target:
            target["area"] = target["area"] * (rw * rh)
        target["size"] = np.asarray((ow, oh), dtype=np.int64)              # the (width, height) pair handed to cv2.resize
        target["im_shape"] = np.asarray(image.shape[:2], dtype=np.int64)
        if "scale_factor" in target:
            target["scale_factor"] *= (rw, rh)
        else:
            target["scale_factor"] = target["size"] / target["orig_size"]
        if "masks" in target:
            raise NotImplementedError("detection Resize: target masks (a training-time transform) are out of scope")
        return target

    def __call__(self, data):
        image, label = data
        oh, ow = self.out_hw(image)
        if np.asarray(image).dtype != np.uint8:
            raise NotImplementedError("detection Resize: uint8 HWC images (as load_image returns them)")
        return _resize_host(image, oh, ow), self.target(image, (oh, ow), label)


def _center_format(b):                                 # corners_to_center_format, transforms.py:304-312
    x0, y0, x1, y1 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], -1)


class Normalize(object):
    """transforms.py:202-235."""

    def __init__(self, mean, std):
        self.mean, self.std = np.asarray(mean, np.float32), np.asarray(std, np.float32)

    def target(self, hw, target):
        if target is None:
            return None
        target = target.copy()
        if "boxes" in target:
            h, w = hw
            target["boxes"] = _center_format(target["boxes"]) / np.asarray([w, h, w, h], dtype=np.float32)
        return target

    def __call__(self, data):
        image, label = data
        out = (np.asarray(image).astype(np.float32) / 255.0 - self.mean) / self.std
        return out, self.target(out.shape[:2], label)


class ToTensor(object):
    """transforms.py:238-247."""

    def __init__(self, data_format="CHW"):
        if data_format not in ("CHW", "HWC"):
            raise ValueError("data_format should be CHW or HWC. Got {}".format(data_format))
        self.data_format = data_format

    def __call__(self, data):
        image, label = data
        if isinstance(image, torch.Tensor):            # already on the device (fused path): layout only
            return (image.permute(2, 0, 1).contiguous() if self.data_format == "CHW" else image), label
        a = np.asarray(image)
        if a.dtype == np.uint8:
            a = a.astype(np.float32) / 255.0
        a = np.ascontiguousarray(np.transpose(a, (2, 0, 1)) if self.data_format == "CHW" else a)
        t = torch.from_numpy(a)
        return (t.cuda() if torch.cuda.is_available() else t), label


class Compose(object):
    """Compose over (image, label) pairs.  [Resize, Normalize] (optionally followed by ToTensor) on a uint8 HWC image runs
    as one device launch when a GPU is present; the labels take the host arithmetic of the two classes."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def _plan(self):
        ts = self.transforms
        if len(ts) >= 2 and isinstance(ts[0], Resize) and isinstance(ts[1], Normalize) and all(isinstance(t, ToTensor) for t in ts[2:]) and len(ts) <= 3:
            return ts[0], ts[1], (ts[2] if len(ts) == 3 else None)
        return None

    def __call__(self, data):
        image, label = data
        plan = self._plan()
        a = image if isinstance(image, np.ndarray) else None
        if plan is not None and a is not None and a.dtype == np.uint8 and a.ndim == 3 and a.shape[-1] in (1, 3) and torch.cuda.is_available():
            rs, nm, tt = plan
            oh, ow = rs.out_hw(a)
            out = device_resize_normalize(torch.from_numpy(np.ascontiguousarray(a)).cuda()[None], (oh, ow), nm.mean, nm.std,
                                          layout=(tt.data_format if tt is not None else "HWC"))[0]
            return out, nm.target((oh, ow), rs.target(a, (oh, ow), label))
        for t in self.transforms:
            data = t(data)
        return data


_tables = {}


def device_resize_normalize(images, out_hw, mean=None, std=None, layout="HWC", dtype=torch.float32):
    """(N, H, W, C) uint8 on the device -> resized to out_hw (cv2 INTER_LINEAR restated), (v / 255 - mean) / std, 'HWC' or
    'CHW', one launch (tlxmi_preprocess_linear_u8)."""
    import ctypes as C
    from .... import _lib, engine as E
    E.need_gpu(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 4:
        raise RuntimeError("device_resize_normalize: a (N, H, W, C) uint8 tensor is expected")
    images = images.contiguous()
    N, H, W, Cc = images.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    tabs = []
    for n_in, n_out in ((W, ow), (H, oh)):
        key = (n_in, n_out, str(images.device))
        if key not in _tables:
            i, w = linear_tables(n_in, n_out)
            _tables[key] = (torch.from_numpy(i).to(images.device), torch.from_numpy(w).to(images.device))
        tabs.append(_tables[key])
    norm = mean is not None

    def per_channel(v):
        t = torch.as_tensor(np.asarray(v, dtype=np.float32)).reshape(-1)
        if t.numel() == 1 and Cc > 1:
            t = t.expand(Cc)
        if t.numel() != Cc:
            raise RuntimeError(f"device_resize_normalize: mean / std must have 1 or {Cc} entries")
        return t.to(images.device).contiguous()
    m = per_channel(mean) if norm else None
    sd = per_channel(std if std is not None else 1.0) if norm else None
    if layout not in ("HWC", "CHW"):
        raise ValueError("layout should be CHW or HWC")
    out = torch.empty((N, Cc, oh, ow) if layout == "CHW" else (N, oh, ow, Cc), dtype=dtype, device=images.device)
    d = _lib.PreprocDesc(N=N, H=H, W=W, C=Cc, out_h=oh, out_w=ow, kh=2, kw=2, out_dtype=E.dt_code(dtype), layout=0 if layout == "CHW" else 1,
                         fold_b=0, cpad=0, normalize=1 if norm else 0)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)      # noqa: E731
    _lib.call("tlxmi_preprocess_linear_u8", C.byref(d), p(images), p(tabs[0][0]), p(tabs[0][1]), p(tabs[1][0]), p(tabs[1][1]), p(m), p(sd),
              p(out), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out
