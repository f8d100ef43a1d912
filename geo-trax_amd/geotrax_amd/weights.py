"""Detector weights: flat tensor files and seeded synthetic weights.

The reference loads an ultralytics ``.pt`` (geotrax/extract.py:222, cfg extraction.model,
geotrax/cfg/default.yaml:81) -- a pickle of ultralytics classes that cannot be read without that
package. This build reads a flat ``.safetensors`` file with the model's ``state_dict`` names
instead (``tools/convert_weights.py`` writes one where ultralytics is installed). Conv+BN pairs
may be stored fused (``model.0.conv.weight/.bias``) or unfused (``.conv.weight`` +
``.bn.{weight,bias,running_mean,running_var}``); unfused pairs are folded here exactly like
ultralytics' ``fuse_conv_and_bn``.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

BN_EPS = 1e-3  # ultralytics sets BatchNorm2d.eps = 1e-3 at model build time

# yolov8.yaml scales: depth, width, max_channels
SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
          "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512)}


def fold_bn(tensors: dict[str, np.ndarray], eps: float = BN_EPS) -> dict[str, np.ndarray]:
    """Folds every ``X.bn.*`` into ``X.conv.{weight,bias}`` (fp64 arithmetic, fp32 result)."""
    out = {}
    for k, v in tensors.items():
        if ".bn." in k or k.endswith("num_batches_tracked"):
            continue
        out[k] = np.asarray(v, dtype=np.float32)
    for k in list(tensors):
        if not k.endswith(".bn.weight"):
            continue
        pfx = k[: -len(".bn.weight")]
        g = tensors[pfx + ".bn.weight"].astype(np.float64)
        b = tensors[pfx + ".bn.bias"].astype(np.float64)
        m = tensors[pfx + ".bn.running_mean"].astype(np.float64)
        var = tensors[pfx + ".bn.running_var"].astype(np.float64)
        w = tensors[pfx + ".conv.weight"].astype(np.float64)
        s = g / np.sqrt(var + eps)
        out[pfx + ".conv.weight"] = (w * s[:, None, None, None]).astype(np.float32)
        cb = tensors.get(pfx + ".conv.bias")
        cb = np.zeros_like(m) if cb is None else cb.astype(np.float64)
        out[pfx + ".conv.bias"] = ((cb - m) * s + b).astype(np.float32)
    return out


def load_weights(path: str | Path) -> dict[str, np.ndarray]:
    from safetensors.numpy import load_file

    t = load_file(str(path))
    t = {k: np.asarray(v, dtype=np.float32) for k, v in t.items()}
    return fold_bn(t)


def save_weights(tensors: dict[str, np.ndarray], path: str | Path) -> None:
    from safetensors.numpy import save_file

    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in tensors.items()}, str(path))


def _make_divisible(x: float, d: int = 8) -> int:
    return int(np.ceil(x / d) * d)


def yolov8_layer_specs(scale: str = "s", nc: int = 4) -> list[tuple[str, tuple[int, ...], bool]]:
    """(tensor name, OIHW shape, has_act) for every conv of a fused YOLOv8 detect model."""
    depth, width, maxc = SCALES[scale]
    ch = lambda c: _make_divisible(min(c, maxc) * width)
    rep = lambda n: max(round(n * depth), 1)
    specs: list[tuple[str, tuple[int, ...], bool]] = []

    def conv(name, cin, cout, k):
        specs.append((name, (cout, cin, k, k), True))

    def c2f(pfx, cin, cout, n):
        c = cout // 2
        conv(f"{pfx}.cv1.conv", cin, 2 * c, 1)
        for k in range(n):
            conv(f"{pfx}.m.{k}.cv1.conv", c, c, 3)
            conv(f"{pfx}.m.{k}.cv2.conv", c, c, 3)
        conv(f"{pfx}.cv2.conv", (2 + n) * c, cout, 1)

    c1, c2, c3, c4, c5 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    conv("model.0.conv", 3, c1, 3)
    conv("model.1.conv", c1, c2, 3)
    c2f("model.2", c2, c2, rep(3))
    conv("model.3.conv", c2, c3, 3)
    c2f("model.4", c3, c3, rep(6))
    conv("model.5.conv", c3, c4, 3)
    c2f("model.6", c4, c4, rep(6))
    conv("model.7.conv", c4, c5, 3)
    c2f("model.8", c5, c5, rep(3))
    conv("model.9.cv1.conv", c5, c5 // 2, 1)
    conv("model.9.cv2.conv", c5 * 2, c5, 1)
    c2f("model.12", c5 + c4, c4, rep(3))
    c2f("model.15", c4 + c3, c3, rep(3))
    conv("model.16.conv", c3, c3, 3)
    c2f("model.18", c3 + c4, c4, rep(3))
    conv("model.19.conv", c4, c4, 3)
    c2f("model.21", c4 + c5, c5, rep(3))
    cb = max(16, c3 // 4, 64)
    cc = max(c3, min(nc, 100))
    for l, cin in enumerate((c3, c4, c5)):
        conv(f"model.22.cv2.{l}.0.conv", cin, cb, 3)
        conv(f"model.22.cv2.{l}.1.conv", cb, cb, 3)
        specs.append((f"model.22.cv2.{l}.2", (64, cb, 1, 1), False))
        conv(f"model.22.cv3.{l}.0.conv", cin, cc, 3)
        conv(f"model.22.cv3.{l}.1.conv", cc, cc, 3)
        specs.append((f"model.22.cv3.{l}.2", (nc, cc, 1, 1), False))
    return specs


def synthetic_yolov8(seed: int = 0, nc: int = 4, scale: str = "s", cls_bias: float = -4.0,
                     gain: float = 1.7, box_decay: float | tuple = 0.3, level_bias: tuple = (0.0, 0.0, 0.0),
                     box_weight_scale: float = 0.3, smooth_cls: bool | int = False) -> dict[str, np.ndarray]:
    """Seeded random fused weights of the YOLOv8 architecture (no checkpoint is reachable here).

    Conv weights ~ N(0, gain^2 / fan_in) so activations keep O(1) scale through the SiLU stack;
    biases ~ N(0, 0.05^2). ``cls_bias`` shifts the class logits so that only a few percent of the
    anchors clear the confidence threshold, which is the load the decode/NMS stage sees on real
    footage. The box branch's final bias decays over the 16 DFL bins
    (-box_decay * bin) and its weights are damped, so decoded boxes span about six strides per
    side -- localised boxes instead of the frame-filling ones uniform DFL logits give. In practice the random
    stack's activations grow with depth (class logits of O(200) at the coarse heads, DFL logits that swamp the
    decaying bias), so with the defaults the strongest anchors sit on the stride-32 head and the boxes come out
    400-700 px wide in a 4K frame, each overlapping ~30 others. The two knobs below put the post-processing in
    the regime of drone footage: ``level_bias`` is added to the class logits of the stride-8/16/32 heads
    ((0, -1e4, -1e4): only stride-8 anchors can fire) and ``box_weight_scale`` scales the last box conv's weights
    (0.002: the decaying bias decides, every side is ~2.9 bins -> boxes of ~46 network pixels, ~90 px in 4K, few
    of which overlap). ``box_decay`` may be a 4-tuple (left, top, right, bottom) for non-square boxes.
    ``smooth_cls`` makes the two 3x3 convs of every class branch and those of the stride-8 neck stage (model.15)
    tap-uniform (each output channel applies one random channel mix to the 3x3 box mean of its input): class logits
    then vary smoothly over neighbouring anchors, so the
    anchors that clear the confidence threshold come in clusters of several per object and NMS has boxes to
    suppress -- the load SURVEY.md 8d.2 asks for (1-3 k candidates in ~132 clusters) instead of isolated ones."""
    rng = np.random.default_rng(seed)
    decay = np.broadcast_to(np.asarray(box_decay, dtype=np.float64), (4,))
    t: dict[str, np.ndarray] = {}
    for name, shape, has_act in yolov8_layer_specs(scale, nc):
        fan_in = shape[1] * shape[2] * shape[3]
        g = gain if has_act else 1.0
        t[name + ".weight"] = (rng.standard_normal(shape) * (g / np.sqrt(fan_in))).astype(np.float32)
        if smooth_cls and shape[2] == 3 and (".cv3." in name or name.startswith("model.15.m.") or
                                             (int(smooth_cls) >= 2 and name.startswith("model.12.m."))):   # 2: the stride-16 neck stage too (twice the radius at stride 8)
            mix = t[name + ".weight"][:, :, 1:2, 1:2] * np.float32(np.sqrt(shape[2] * shape[3]))   # keeps the output variance for smooth inputs
            t[name + ".weight"] = np.broadcast_to(mix / np.float32(shape[2] * shape[3]), shape).astype(np.float32).copy()
        b = rng.standard_normal(shape[0]) * 0.05
        if name.endswith("cv3.0.2") or name.endswith("cv3.1.2") or name.endswith("cv3.2.2"):
            b = b + cls_bias + float(level_bias[int(name.split(".")[3])])
        if ".cv2." in name and name.endswith(".2"):
            t[name + ".weight"] *= np.float32(box_weight_scale)
            b = b - np.repeat(decay, 16) * np.tile(np.arange(16), 4)
        t[name + ".bias"] = b.astype(np.float32)
    return t


def calibrate_cls_bias(tensors: dict[str, np.ndarray], raw_logits: np.ndarray, conf: float, target: int) -> dict[str, np.ndarray]:
    """Returns a copy of `tensors` whose class-logit biases are shifted by one constant so that
    about `target` anchors of the probed frame clear `conf`. raw_logits: [anchors, nc] class
    logits of one forward pass with the unshifted weights (Detector.raw_output(logits=True)[:, 4:])."""
    logit = np.sort(raw_logits.max(1).astype(np.float64))[::-1]
    k = min(max(int(target), 1), len(logit) - 1)
    delta = np.log(conf / (1 - conf)) - 0.5 * (logit[k - 1] + logit[k])
    out = dict(tensors)
    for name in tensors:
        if ".cv3." in name and name.endswith(".2.bias"):
            out[name] = (tensors[name] + np.float32(delta)).astype(np.float32)
    return out
