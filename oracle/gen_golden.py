"""Generate tests/golden/*.npz — run ONLY in the development container (needs /root/reference).

For each torch-capable reference model file it
  1. imports the file UNMODIFIED from /root/reference by path, with `tensorlayerx` resolved to the
     oracle's torch-CPU stand-in (oracle/tlx_cpu) — the reference source is never copied;
  2. fills it and the restatement (oracle/functional.py) from the same seeded numpy recipe;
  3. requires restatement == reference graph (max abs diff <= 1e-5, same argmax);
  4. writes a small fixture: recipe ids + expected fp32 logits + argmax.
The Paddle-converted files (swin_transformer.py, mobilenetv2.py, mobilenetv3.py) and detection/yolov3.py are loaded the
same way — unmodified, by path — with the packages they hard-import and this image lacks (paddle, paddle2tlx,
decorator, torchvision) resolved to the small import shims of oracle/shims/ (README there lists what each provides).

    python -m oracle.gen_golden            # writes every fixture of tests/golden/
    python -m oracle.gen_golden swin yolo  # only the fixtures whose file name contains one of the words
"""
import importlib
import importlib.util
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

from oracle import functional as OF  # noqa: E402
from tlxcv_amd import seeded  # noqa: E402


SHIMS = os.path.join(REPO, "oracle", "shims")
_TLX_KEYS = ("tensorlayerx", "tensorlayerx.nn", "tensorlayerx.ops", "tensorlayerx.nn.initializers", "tensorlayerx.initializers")
_SHIM_KEYS = ("paddle", "paddle2tlx", "paddle2tlx.pd2tlx", "paddle2tlx.pd2tlx.ops", "paddle2tlx.pd2tlx.ops.tlxops",
              "paddle2tlx.pd2tlx.utils", "decorator", "torchvision", "torchvision.ops",
              # mobilenetv2.py:6-7 / mobilenetv3.py:10-11 import their siblings as TOP-LEVEL packages (`from utils.common_func
              # import ...`, `from ops.ops_fusion import ...`): they resolve with the classification directory on sys.path
              "utils", "utils.common_func", "ops", "ops.ops_fusion", "ops.theseus_layer")


def import_reference(relpath, modname, paddle=False, package=None):
    """Load one reference source file UNMODIFIED, by path, with `tensorlayerx` resolved to the oracle's torch-CPU
    stand-in (oracle/tlx_cpu).  paddle=True: the file is one of the Paddle-converted ones — `paddle`, `paddle2tlx`,
    `decorator`, `torchvision` resolve to oracle/shims/ (see its README) and the functions of the stand-in return
    PdTensor (Paddle tensor-method spellings, oracle/tlx_cpu/pd.py).  package=(name, dir): load the file as module
    `name.<stem>` of a synthetic parent package rooted at `dir`, so its relative imports resolve against the reference
    tree without running the parent's __init__ (detection/__init__.py pulls detr / ssd / ppyoloe)."""
    import types
    import oracle.tlx_cpu as tlx_cpu
    from oracle.tlx_cpu import pd
    saved = {k: sys.modules.get(k) for k in _TLX_KEYS + _SHIM_KEYS}
    saved_path = list(sys.path)
    tlx = tlx_cpu
    ops = tlx_cpu.ops
    if paddle:
        ops = pd.pd_module(tlx_cpu.ops, "tensorlayerx.ops")
        tlx = pd.pd_module(tlx_cpu, "tensorlayerx", ops=ops, arange=ops.arange, stack=ops.stack)
    for k in _SHIM_KEYS:
        sys.modules.pop(k, None)
    sys.modules["tensorlayerx"] = tlx
    sys.modules["tensorlayerx.nn"] = tlx_cpu.nn
    sys.modules["tensorlayerx.ops"] = ops
    sys.modules["tensorlayerx.nn.initializers"] = tlx_cpu.nn.initializers
    sys.modules["tensorlayerx.initializers"] = tlx_cpu.nn.initializers
    sys.path.insert(0, SHIMS)
    sys.path.insert(0, os.path.join(REF, "tlxcv", "models", "classification"))
    created = []
    try:
        if package is not None:
            pname, pdir = package
            parent = types.ModuleType(pname)
            parent.__path__ = [os.path.join(REF, pdir)]
            parent.__package__ = pname
            sys.modules[pname] = parent
            created.append(pname)
            stem = os.path.splitext(os.path.basename(relpath))[0]
            mod = importlib.import_module(f"{pname}.{stem}")
            created += [k for k in sys.modules if k.startswith(pname + ".")]
            return mod
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        sys.path[:] = saved_path
        for k in created:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _check(name, ref_out, re_out):
    d = (ref_out - re_out).abs().max().item()
    same = bool((ref_out.argmax(-1) == re_out.argmax(-1)).all()) if ref_out.dim() == 2 else True
    print(f"[{name}] reference-file vs restatement: max|diff| = {d:.3e}, argmax equal = {same}")
    assert d <= 1e-5 and same, f"{name}: restatement disagrees with the reference graph"
    return d


def gen_resnet(depth, batch, wseed, xseed, fname):
    ref = import_reference("tlxcv/models/classification/resnet.py", "ref_resnet")
    model = getattr(ref, f"resnet{depth}")()
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed))
    torch.manual_seed(0)
    with torch.no_grad():
        ref_out = model(x)
        re_out = OF.resnet({k: torch.from_numpy(v) for k, v in params.items()}, x, depth)
    d = _check(f"resnet{depth}", ref_out, re_out)
    np.savez_compressed(
        os.path.join(OUT, fname), arch=f"resnet{depth}", weight_seed=wseed, input_seed=xseed, batch=batch,
        logits=ref_out.numpy().astype(np.float32), argmax=ref_out.argmax(-1).numpy().astype(np.int64),
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_vit(arch, batch, wseed, xseed, fname, hw=224):
    ref = import_reference("tlxcv/models/classification/vision_transformer.py", "ref_vit")
    model = getattr(ref, arch)()
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed, hw=hw))
    with torch.no_grad():
        ref_out = model(x)
        re_out = OF.vit({k: torch.from_numpy(v) for k, v in params.items()}, x, arch)
    d = _check(arch, ref_out, re_out)
    np.savez_compressed(
        os.path.join(OUT, fname), arch=arch, weight_seed=wseed, input_seed=xseed, batch=batch, hw=hw,
        logits=ref_out.numpy().astype(np.float32), argmax=ref_out.argmax(-1).numpy().astype(np.int64),
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_mobilenetv1(batch, wseed, xseed, fname):
    ref = import_reference("tlxcv/models/classification/mobilenetv1.py", "ref_mobilenetv1")
    model = ref.MobileNetV1()
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed))
    with torch.no_grad():
        ref_out = model(x)
        re_out = OF.mobilenetv1({k: torch.from_numpy(v) for k, v in params.items()}, x)
    d = _check("mobilenetv1", ref_out, re_out)
    np.savez_compressed(
        os.path.join(OUT, fname), arch="MobileNetV1", weight_seed=wseed, input_seed=xseed, batch=batch,
        logits=ref_out.numpy().astype(np.float32), argmax=ref_out.argmax(-1).numpy().astype(np.int64),
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_vgg(arch, batch_norm, batch, wseed, xseed, fname):
    ref = import_reference("tlxcv/models/classification/vgg.py", "ref_vgg")
    model = getattr(ref, arch)(batch_norm=batch_norm)
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed))
    with torch.no_grad():
        ref_out = model(x)
        re_out = OF.vgg({k: torch.from_numpy(v) for k, v in params.items()}, x, arch, batch_norm)
    d = _check(arch + ("_bn" if batch_norm else ""), ref_out, re_out)
    np.savez_compressed(
        os.path.join(OUT, fname), arch=arch, batch_norm=batch_norm, weight_seed=wseed, input_seed=xseed, batch=batch,
        logits=ref_out.numpy().astype(np.float32), argmax=ref_out.argmax(-1).numpy().astype(np.int64),
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_resnext(layers, cardinality, batch, hw, wseed, xseed, fname):
    ref = import_reference("tlxcv/models/classification/resnext.py", "ref_resnext")
    model = ref.ResNeXt(layers=layers, cardinality=cardinality)
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed, hw))
    with torch.no_grad():
        ref_out = model(x)
        re_out = OF.resnext({k: torch.from_numpy(v) for k, v in params.items()}, x, layers, cardinality)
    d = _check(f"resnext{layers}_{cardinality}x4d", ref_out, re_out)
    np.savez_compressed(
        os.path.join(OUT, fname), layers=layers, cardinality=cardinality, weight_seed=wseed, input_seed=xseed, batch=batch,
        hw=hw, logits=ref_out.numpy().astype(np.float32), argmax=ref_out.argmax(-1).numpy().astype(np.int64),
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_efficientnet(arch, batch, hw, wseed, xseed, fname):
    ref = import_reference("tlxcv/models/classification/efficientnet.py", "ref_efficientnet")
    model = ref.efficientnet(arch)           # builds itself by one forward of ones (efficientnet.py:433-441)
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed, hw))
    with torch.no_grad():
        ref_out = model(x)
        re_out = OF.efficientnet({k: torch.from_numpy(v) for k, v in params.items()}, x, arch)
    d = _check(arch, ref_out, re_out)
    np.savez_compressed(
        os.path.join(OUT, fname), arch=arch, weight_seed=wseed, input_seed=xseed, batch=batch, hw=hw,
        logits=ref_out.numpy().astype(np.float32), argmax=ref_out.argmax(-1).numpy().astype(np.int64),
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_resnest(arch, batch, hw, wseed, xseed, fname):
    ref = import_reference("tlxcv/models/classification/resnest.py", "ref_resnest")
    model = getattr(ref, arch)()
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed, hw))
    with torch.no_grad():
        ref_out = model(x)
        re_out = OF.resnest({k: torch.from_numpy(v) for k, v in params.items()}, x, arch)
    d = _check(arch, ref_out, re_out)
    np.savez_compressed(
        os.path.join(OUT, fname), arch=arch, weight_seed=wseed, input_seed=xseed, batch=batch, hw=hw,
        logits=ref_out.numpy().astype(np.float32), argmax=ref_out.argmax(-1).numpy().astype(np.int64),
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_alexnet(batch, wseed, xseed, fname):
    ref = import_reference("tlxcv/models/classification/alexnet.py", "ref_alexnet")
    model = ref.alexnet()
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed))
    with torch.no_grad():
        ref_out = model(x)
        re_out = OF.alexnet({k: torch.from_numpy(v) for k, v in params.items()}, x)
    d = _check("alexnet", ref_out, re_out)
    np.savez_compressed(
        os.path.join(OUT, fname), arch="alexnet", weight_seed=wseed, input_seed=xseed, batch=batch,
        logits=ref_out.numpy().astype(np.float32), argmax=ref_out.argmax(-1).numpy().astype(np.int64),
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_darknet(batch, hw, wseed, xseed, fname):
    ref = import_reference("tlxcv/models/detection/backbones/darknet.py", "ref_darknet")
    model = ref.DarkNet()
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed, hw=hw))
    with torch.no_grad():
        ref_out = model({"images": x})
        re_out = OF.darknet53({k: torch.from_numpy(v) for k, v in params.items()}, x)
    d = max(_check(f"darknet53 stage {i}", a, b) for i, (a, b) in enumerate(zip(ref_out, re_out)))
    np.savez_compressed(
        os.path.join(OUT, fname), arch="DarkNet53", weight_seed=wseed, input_seed=xseed, batch=batch, hw=hw,
        feat0=ref_out[0].numpy(), feat1=ref_out[1].numpy(), feat2=ref_out[2].numpy(),
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_mobilenet_det(batch, hw, wseed, xseed, fname):
    """The detection MobileNet backbone (backbones/mobilenet_v1.py), with the SSD-style extra blocks switched on so that
    ExtraBlock / relu6 are covered: feature maps after blocks 4, 6, 13 and the first two extra blocks."""
    ref = import_reference("tlxcv/models/detection/backbones/mobilenet_v1.py", "ref_mobilenet_det")
    kw = dict(feature_maps=[4, 6, 13, 14, 15], with_extra_blocks=True, extra_block_filters=[[256, 512], [128, 256]])
    model = ref.MobileNet(**kw)
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed, hw=hw))
    with torch.no_grad():
        ref_out = model({"images": x})
        re_out = OF.mobilenet_det({k: torch.from_numpy(v) for k, v in params.items()}, x, feature_maps=kw["feature_maps"],
                                  extra_block_filters=kw["extra_block_filters"])
    assert len(ref_out) == len(re_out) == 5
    d = max(_check(f"mobilenet_det feature {i}", a, b) for i, (a, b) in enumerate(zip(ref_out, re_out)))
    np.savez_compressed(
        os.path.join(OUT, fname), arch="MobileNet(det)", weight_seed=wseed, input_seed=xseed, batch=batch, hw=hw,
        **{f"feat{i}": o.numpy() for i, o in enumerate(ref_out)},
        restatement_max_abs_diff=np.float64(d), pinned_by="reference-file-on-tlx_cpu",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_detr_mha(fname):
    """DETR's MultiHeadAttention (detection/detr.py:965-1062), the reference's own class loaded by path: a cross-attention
    case (7 queries over 13 memory tokens, batch 2, with an additive mask, weights wanted) and a self-attention case
    (197 tokens, the ViT length, no mask)."""
    ref = import_reference("tlxcv/models/detection/detr.py", "ref_detr")
    out = {}
    rng = np.random.default_rng(40)
    for tag, D, H, T, S, B, masked in (("cross", 64, 4, 7, 13, 2, True), ("self", 128, 4, 197, 197, 2, False)):
        m = ref.MultiHeadAttention(D, H)
        shapes = seeded.shapes_of(m)
        params = seeded.fill(shapes, 41)
        m.load_dict(params)
        m.set_eval()
        q = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32))
        kv = q if tag == "self" else torch.from_numpy(rng.standard_normal((S, B, D)).astype(np.float32))
        mask = None
        if masked:
            mask = torch.from_numpy(np.where(rng.random((T, S)) < 0.25, -1e9, 0.0).astype(np.float32))
            mask[:, 0] = 0.0
        with torch.no_grad():
            o, w = m((q, kv, kv), attn_mask=mask)
            o2, w2 = OF.detr_mha({k: torch.from_numpy(v) for k, v in params.items()}, "", q, kv, kv, H, mask)
        d = max(_check(f"detr mha {tag} out", o.reshape(-1, D), o2.reshape(-1, D)), _check(f"detr mha {tag} weights", w.reshape(-1, S), w2.reshape(-1, S)))
        out.update({f"{tag}_q": q.numpy(), f"{tag}_kv": kv.numpy(), f"{tag}_out": o.numpy(), f"{tag}_weights": w.numpy(),
                    f"{tag}_dims": np.array([D, H, T, S, B]), f"{tag}_diff": np.float64(d)})
        if mask is not None:
            out[f"{tag}_mask"] = mask.numpy()
    np.savez_compressed(os.path.join(OUT, fname), weight_seed=41, pinned_by="reference-file-on-tlx_cpu",
                        restatement_max_abs_diff=np.float64(max(out["cross_diff"], out["self_diff"])),
                        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__, **out)


def gen_yolov3(batch, hw, wseed, xseed, fname):
    """detection/yolov3.py loaded unmodified as a module of a synthetic parent package (its relative imports resolve
    against the reference tree; `decorator` / `torchvision` from oracle/shims).  YOLOv3.forward (yolov3.py:51-104) is
    followed by hand up to the head outputs — backbone, neck, yolo_head of the reference's own model object — because
    the next line, post_process, needs `yolo_box_func`, which is None off Paddle (detection/utils/ops.py:436-452)."""
    ref = import_reference("tlxcv/models/detection/yolov3.py", "ref_yolov3", package=("refdet", "tlxcv/models/detection"))
    model = ref.YOLOv3()
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed, hw=hw))
    with torch.no_grad():
        body = model.backbone({"images": x})
        neck = model.neck(body, model.for_mot)
        head = model.yolo_head({"images": x, "neck_feats": neck})
        _, neck_re, head_re = OF.yolov3({k: torch.from_numpy(v) for k, v in params.items()}, x)
    d = max([_check(f"yolov3 head {i}", a, b) for i, (a, b) in enumerate(zip(head, head_re))]
            + [_check(f"yolov3 neck {i}", a, b) for i, (a, b) in enumerate(zip(neck, neck_re))])
    np.savez_compressed(
        os.path.join(OUT, fname), arch="YOLOv3", weight_seed=wseed, input_seed=xseed, batch=batch, hw=hw,
        head0=head[0].numpy(), head1=head[1].numpy(), head2=head[2].numpy(), neck2=neck[2].numpy(),
        restatement_max_abs_diff=np.float64(d),
        pinned_by="reference-file-on-tlx_cpu (backbone, neck, yolo_head of the reference's YOLOv3 object; decorator / "
                  "torchvision from oracle/shims)",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_paddle_converted(relpath, ctor_name, fn, batch, wseed, xseed, fname, hw=224):
    """A Paddle-converted reference classifier (swin_transformer.py, mobilenetv2.py, mobilenetv3.py) loaded unmodified:
    paddle / paddle2tlx from oracle/shims, Paddle tensor-method spellings from oracle/tlx_cpu/pd.py."""
    from oracle.tlx_cpu import pd
    ref = import_reference(relpath, "ref_" + ctor_name, paddle=True)
    model = getattr(ref, ctor_name)()
    shapes = seeded.shapes_of(model)
    params = seeded.fill(shapes, wseed)
    model.load_dict(params)
    model.set_eval()
    x = torch.from_numpy(seeded.image_batch(batch, xseed, hw=hw))
    with torch.no_grad():
        ref_out = pd.unwrap(model(pd.wrap(x)))
        re_out = fn({k: torch.from_numpy(v) for k, v in params.items()}, x)
    d = _check(ctor_name, ref_out, re_out)
    np.savez_compressed(
        os.path.join(OUT, fname), arch=ctor_name, weight_seed=wseed, input_seed=xseed, batch=batch, hw=hw,
        logits=ref_out.numpy().astype(np.float32), argmax=ref_out.argmax(-1).numpy().astype(np.int64),
        restatement_max_abs_diff=np.float64(d),
        pinned_by="reference-file-on-tlx_cpu (paddle / paddle2tlx import shims, Paddle tensor-method spellings: oracle/shims, oracle/tlx_cpu/pd.py)",
        param_names=np.array(list(shapes.keys())), torch_version=torch.__version__)


def gen_yolo_post(fname):
    """YOLOv3 post-processing on the yolov3_b1 case: head maps from the reference's own model object;
    decode = oracle/detection.py's restatement of Paddle's yolo_box (UNPINNED: the op exists on the Paddle backend only);
    NMS = the reference's OWN tlx_multiclass_nms (detection/utils/ops.py:255-329, torchvision.ops from oracle/shims) run on
    those boxes, and required to equal the restatement.  Plus a denser synthetic NMS case (overlapping boxes, 7 classes)."""
    from oracle import detection as OD
    ref = import_reference("tlxcv/models/detection/yolov3.py", "ref_yolov3_post", package=("refdet", "tlxcv/models/detection"))
    ref_nms = ref.cvt_results.__globals__["tlx_multiclass_nms"]
    g = np.load(os.path.join(OUT, "yolov3_b1.npz"))
    heads = [torch.from_numpy(g[f"head{i}"]) for i in range(3)]
    model = ref.YOLOv3()
    anchors = model.yolo_head.mask_anchors
    hw = int(g["hw"])
    im_shape = torch.tensor([[hw, hw]], dtype=torch.float32)
    boxes, scores = OD.yolo_decode(heads, anchors, 92, im_shape, torch.ones_like(im_shape), conf_thresh=0.005, downsample_ratio=32)
    sys.path.insert(0, SHIMS)
    try:
        out = {}
        for tag, b, s_, thr in (("yolo", boxes, scores, 0.01), ("dense", None, None, 0.3)):
            if b is None:
                rng = np.random.default_rng(50)
                ctr, wh = rng.uniform(0, 200, (2, 600, 2)), rng.uniform(5, 60, (2, 600, 2))
                b = torch.from_numpy(np.concatenate([ctr - wh / 2, ctr + wh / 2], -1).astype(np.float32))
                s_ = torch.from_numpy((rng.random((2, 600, 7)) ** 3).astype(np.float32))
            det_ref = ref_nms(b, s_, score_threshold=thr, nms_threshold=0.5, keep_top_k=100)
            det_re = OD.multiclass_nms(b, s_, thr, 0.5, 100)
            for x, y in zip(det_ref, det_re):
                assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), "NMS restatement disagrees with tlx_multiclass_nms"
            out[f"{tag}_boxes"], out[f"{tag}_scores"] = b.numpy(), s_.numpy()
            out[f"{tag}_counts"] = np.array([0 if d is None else d.shape[0] for d in det_ref], dtype=np.int32)
            out[f"{tag}_det"] = np.concatenate([np.zeros((0, 6), np.float32)] + [d.numpy() for d in det_ref if d is not None])
            out[f"{tag}_thr"] = np.float32(thr)
            print(f"[yolo post {tag}] {b.shape[1]} boxes -> detections per image {out[f'{tag}_counts'].tolist()}")
    finally:
        sys.path.remove(SHIMS)
    np.savez_compressed(os.path.join(OUT, fname), anchors=np.array(anchors, dtype=np.float32), hw=hw,
                        pinned_by="reference-file-on-tlx_cpu (NMS: the reference's tlx_multiclass_nms with torchvision.ops from oracle/shims; "
                                  "box decode: restatement of paddle.vision.ops.yolo_box, UNPINNED — Paddle-only in the reference)",
                        restatement_max_abs_diff=np.float64(0.0), **out)


def gen_tlx_npz(fname):
    """SURVEY 8f rank 1 — checkpoint interchange.  Two small instances of the reference's own classes (its
    VisionTransformer at 32 x 32 / width 32 / depth 2, its MobileNetV1 at scale 0.125, 10 classes) are filled from the
    seeded recipe on the stand-in and written with `model.save_weights(path)` (train.py:55): the positional `params`
    object array.  The fixture keeps that array, the seeds, and the reference models' logits on a seeded input, so the
    engine side can be checked to (i) restore every named weight from the positional list and (ii) compute the same
    logits from the restored weights.  The on-disk format itself is [TLX-recalled] (oracle/tlx_cpu/nn.py)."""
    import tempfile
    out = {}
    vit = import_reference("tlxcv/models/classification/vision_transformer.py", "ref_vit_small_instance")
    mb = import_reference("tlxcv/models/classification/mobilenetv1.py", "ref_mobilenetv1_small_instance")
    cases = (("vit", lambda: vit.VisionTransformer(img_size=32, patch_size=8, num_classes=10, embed_dim=32, depth=2, num_heads=2,
                                                   mlp_ratio=2, qkv_bias=True, epsilon=1e-6), 21, 32),
             ("mbv1", lambda: mb.MobileNetV1(scale=0.125, num_classes=10), 22, 64))
    for tag, ctor, wseed, hw in cases:
        model = ctor()
        shapes = seeded.shapes_of(model)
        model.load_dict(seeded.fill(shapes, wseed))
        model.set_eval()
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "model.npz")
            model.save_weights(path)
            fresh = ctor()
            fresh.load_weights(path)
            fresh.set_eval()
            params = np.load(path, allow_pickle=True)["params"]
        x = torch.from_numpy(seeded.image_batch(2, 30, hw=hw))
        with torch.no_grad():
            y, y2 = model(x), fresh(x)
        assert torch.equal(y, y2), "stand-in save_weights -> load_weights must round-trip"
        print(f"[tlx npz {tag}] {len(params)} arrays, {sum(int(np.prod(a.shape)) for a in params)} values, logits std {y.std().item():.3f}")
        for i, a in enumerate(params):
            out[f"{tag}_params_{i:03d}"] = np.asarray(a, dtype=np.float32)
        out[f"{tag}_n"] = len(params)
        out[f"{tag}_weight_seed"], out[f"{tag}_input_seed"], out[f"{tag}_hw"] = wseed, 30, hw
        out[f"{tag}_logits"] = y.numpy().astype(np.float32)
        out[f"{tag}_param_names"] = np.array(list(shapes.keys()))
    np.savez_compressed(os.path.join(OUT, fname), pinned_by="reference-file-on-tlx_cpu (small instances of the reference's VisionTransformer / MobileNetV1; "
                        "positional checkpoint written by the stand-in's save_weights, format [TLX-recalled])",
                        restatement_max_abs_diff=np.float64(0.0), **out)


def main(only=()):
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    jobs = []

    def job(fn, *args, **kw):
        fname = next(a for a in args if isinstance(a, str) and a.endswith('.npz'))
        if not only or any(w in fname for w in only):
            jobs.append((fn, args, kw))

    job(gen_resnet, 50, 4, 1, 0, "resnet50_b4.npz")      # BASELINE.json configs[0]
    job(gen_resnet, 18, 2, 11, 10, "resnet18_b2.npz")
    job(gen_vit, "vit_base_patch16_224", 2, 2, 0, "vit_b16_b2.npz")       # BASELINE.json configs[2] graph
    job(gen_vit, "vit_small_patch16_224", 1, 12, 3, "vit_small_b1.npz")   # no qkv bias, qk_scale override, hd=96
    job(gen_vit, "vit_base_patch16_384", 1, 21, 19, "vit_b16_384_b1.npz", hw=384)      # 577 tokens: the long-sequence attention
    swin = "tlxcv/models/classification/swin_transformer.py"
    job(gen_paddle_converted, swin, "swintransformer_base_patch4_window7_224",
                         lambda p, x: OF.swin(p, x, "swintransformer_base_patch4_window7_224"), 2, 3, 0, "swin_b_b2.npz")   # BASELINE.json configs[3] graph
    job(gen_paddle_converted, swin, "swintransformer_tiny_patch4_window7_224",
                         lambda p, x: OF.swin(p, x, "swintransformer_tiny_patch4_window7_224"), 1, 13, 4, "swin_t_b1.npz")
    job(gen_paddle_converted, swin, "swintransformer_base_patch4_window12_384",            # 144-token windows, 384 x 384
                         lambda p, x: OF.swin(p, x, "swintransformer_base_patch4_window12_384"), 1, 22, 20, "swin_b_w12_384_b1.npz", hw=384)
    job(gen_paddle_converted, "tlxcv/models/classification/mobilenetv2.py", "mobilenet_v2", lambda p, x: OF.mobilenetv2(p, x),
                         2, 7, 5, "mobilenetv2_b2.npz", hw=128)
    mbv3 = "tlxcv/models/classification/mobilenetv3.py"
    job(gen_paddle_converted, mbv3, "mobilenet_v3_small", lambda p, x: OF.mobilenetv3(p, x, OF.MBV3_SMALL), 2, 8, 6,
                         "mobilenetv3_small_b2.npz", hw=128)
    job(gen_paddle_converted, mbv3, "mobilenet_v3_large", lambda p, x: OF.mobilenetv3(p, x, OF.MBV3_LARGE), 1, 9, 7,
                         "mobilenetv3_large_b1.npz", hw=128)
    job(gen_mobilenetv1, 2, 4, 1, "mobilenetv1_b2.npz")
    job(gen_vgg, "vgg16", False, 1, 10, 8, "vgg16_b1.npz")
    job(gen_vgg, "vgg11", True, 2, 11, 9, "vgg11_bn_b2.npz")
    job(gen_alexnet, 2, 12, 10, "alexnet_b2.npz")
    job(gen_resnext, 50, 32, 2, 96, 13, 11, "resnext50_32x4d_b2.npz")
    job(gen_resnext, 50, 64, 1, 64, 14, 12, "resnext50_64x4d_b1.npz")
    job(gen_resnest, "resnest50", 2, 96, 17, 15, "resnest50_b2.npz")
    job(gen_resnest, "resnest50_fast_1s1x64d", 1, 64, 18, 16, "resnest50_fast_b1.npz")     # radix 1 (sigmoid gate), avd_first
    job(gen_efficientnet, "efficientnet_b0", 2, 224, 15, 13, "efficientnet_b0_b2.npz")
    job(gen_efficientnet, "efficientnet_b2", 1, 130, 16, 14, "efficientnet_b2_b1.npz")     # width / depth multipliers, odd extents under 'SAME' 
    job(gen_darknet, 1, 64, 5, 2, "darknet53_b1.npz")
    job(gen_yolov3, 1, 64, 6, 3, "yolov3_b1.npz")
    job(gen_mobilenet_det, 1, 128, 23, 17, "mobilenet_det_b1.npz")
    job(gen_tlx_npz, "tlx_npz_small.npz")
    job(gen_detr_mha, "detr_mha.npz")
    job(gen_yolo_post, "yolov3_post_b1.npz")
    for fn, args, kw in jobs:
        fn(*args, **kw)


if __name__ == "__main__":
    main(tuple(sys.argv[1:]))
