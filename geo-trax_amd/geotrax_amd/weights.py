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
    t = fold_bn(t)
    if is_rtdetr(t):
        t = fold_input_proj(fuse_repconv(t))
    return t


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


# --------------------------------------------------------------------------- RT-DETR (rtdetr-l topology)
# The reference swaps YOLO for RTDETR when the model's yaml says so (geotrax/extract.py:222-225). Tensor names are ultralytics'
# state_dict names of cfg/models/rt-detr/rtdetr-l.yaml; load_weights() brings a checkpoint to the fused form the library and the
# oracle read: Conv+BN folded (fold_bn), RepConv's 3x3 + 1x1 pair fused into `.conv` (fuse_repconv), the decoder's
# Sequential(Conv2d, BatchNorm2d) input projections folded into `.0.weight` / `.0.bias` (fold_input_proj).

def is_rtdetr(tensors: dict) -> bool:
    return any(k.startswith("model.28.decoder.layers.") for k in tensors)


def fuse_repconv(t: dict[str, np.ndarray]) -> dict[str, np.ndarray]:
    """RepConv.fuse_convs on folded tensors: `X.conv1.conv` (3x3) + `X.conv2.conv` (1x1, same Cin) -> `X.conv` (3x3, the 1x1 kernel
    added at the centre tap; biases summed). LightConv's pair (1x1 then depthwise) has other shapes and is left alone."""
    out = dict(t)
    for k in list(t):
        if not k.endswith(".conv1.conv.weight"):
            continue
        p = k[: -len(".conv1.conv.weight")]
        w3, w1 = t[k], t.get(p + ".conv2.conv.weight")
        if w1 is None or w3.shape[2] != 3 or w1.shape[2] != 1 or w1.shape[1] != w3.shape[1] or w3.shape[1] == 1:
            continue
        w = w3.astype(np.float64).copy()
        w[:, :, 1, 1] += w1[:, :, 0, 0].astype(np.float64)
        b = t.get(p + ".conv1.conv.bias", 0).astype(np.float64) + t.get(p + ".conv2.conv.bias", 0).astype(np.float64)
        for s in (".conv1.conv.weight", ".conv1.conv.bias", ".conv2.conv.weight", ".conv2.conv.bias"):
            out.pop(p + s, None)
        out[p + ".conv.weight"], out[p + ".conv.bias"] = w.astype(np.float32), b.astype(np.float32)
    return out


def fold_input_proj(t: dict[str, np.ndarray], eps: float = BN_EPS) -> dict[str, np.ndarray]:
    out = dict(t)
    for k in list(t):
        if ".input_proj." not in k or not k.endswith(".1.running_var"):
            continue
        p = k[: -len(".1.running_var")]
        s = t[p + ".1.weight"].astype(np.float64) / np.sqrt(t[k].astype(np.float64) + eps)
        out[p + ".0.weight"] = (t[p + ".0.weight"].astype(np.float64) * s[:, None, None, None]).astype(np.float32)
        out[p + ".0.bias"] = (t[p + ".1.bias"].astype(np.float64) - t[p + ".1.running_mean"].astype(np.float64) * s).astype(np.float32)
        for q in (".1.weight", ".1.bias", ".1.running_mean", ".1.running_var", ".1.num_batches_tracked"):
            out.pop(p + q, None)
    return out


def rtdetr_layer_specs(nc: int = 80, width: float = 1.0, hd: int = 256, ndl: int = 6, nh: int = 8, npts: int = 4, d_ffn: int = 1024):
    """(name, shape, kind) for every tensor of a fused RT-DETR-l; kind: 'relu' / 'silu' / 'lin' conv or linear weights (with a
    bias of the leading dim), 'ln' LayerNorm pair. width scales the backbone / encoder channel counts (multiples of 16)."""
    ch = lambda c: max(16, int(round(c * width / 16)) * 16)
    specs = []

    def conv(name, cin, cout, k, kind, groups=1):
        specs.append((name, (cout, cin // groups, k, k), kind))

    def lin(name, cin, cout):
        specs.append((name, (cout, cin), "lin"))

    def ln(name, c):
        specs.append((name, (c,), "ln"))

    cm = ch(32)
    conv("model.0.stem1.conv", 3, cm, 3, "relu")
    conv("model.0.stem2a.conv", cm, max(8, cm // 2), 2, "relu")
    conv("model.0.stem2b.conv", max(8, cm // 2), cm, 2, "relu")
    conv("model.0.stem3.conv", 2 * cm, cm, 3, "relu")
    conv("model.0.stem4.conv", cm, ch(48), 1, "relu")

    def hgblock(p, c1, cmid, c2, k, light, n=6):
        for i in range(n):
            cin = c1 if i == 0 else cmid
            if light:
                conv(f"{p}.m.{i}.conv1.conv", cin, cmid, 1, "lin")
                conv(f"{p}.m.{i}.conv2.conv", cmid, cmid, k, "relu", groups=cmid)
            else:
                conv(f"{p}.m.{i}.conv", cin, cmid, k, "relu")
        conv(p + ".sc.conv", c1 + n * cmid, c2 // 2, 1, "relu")
        conv(p + ".ec.conv", c2 // 2, c2, 1, "relu")

    c1, c2, c3, c4 = ch(128), ch(512), ch(1024), ch(2048)
    hgblock("model.1", ch(48), ch(48), c1, 3, False)
    conv("model.2.conv", c1, c1, 3, "lin", groups=c1)
    hgblock("model.3", c1, ch(96), c2, 3, False)
    conv("model.4.conv", c2, c2, 3, "lin", groups=c2)
    hgblock("model.5", c2, ch(192), c3, 5, True)
    hgblock("model.6", c3, ch(192), c3, 5, True)
    hgblock("model.7", c3, ch(192), c3, 5, True)
    conv("model.8.conv", c3, c3, 3, "lin", groups=c3)
    hgblock("model.9", c3, ch(384), c4, 5, True)
    e = hd
    conv("model.10.conv", c4, e, 1, "lin")
    lin("model.11.ma.in_proj", e, 3 * e)
    lin("model.11.ma.out_proj", e, e)
    lin("model.11.fc1", e, d_ffn)
    lin("model.11.fc2", d_ffn, e)
    ln("model.11.norm1", e)
    ln("model.11.norm2", e)

    def repc3(p, cin, c):
        conv(p + ".cv1.conv", cin, c, 1, "silu")
        conv(p + ".cv2.conv", cin, c, 1, "silu")
        for i in range(3):
            conv(f"{p}.m.{i}.conv", c, c, 3, "silu")

    conv("model.12.conv", e, e, 1, "silu")
    conv("model.14.conv", c3, e, 1, "lin")
    repc3("model.16", 2 * e, e)
    conv("model.17.conv", e, e, 1, "silu")
    conv("model.19.conv", c2, e, 1, "lin")
    repc3("model.21", 2 * e, e)
    conv("model.22.conv", e, e, 3, "silu")
    repc3("model.24", 2 * e, e)
    conv("model.25.conv", e, e, 3, "silu")
    repc3("model.27", 2 * e, e)
    d = "model.28"
    for i in range(3):
        conv(f"{d}.input_proj.{i}.0", e, hd, 1, "lin")
    for i in range(ndl):
        lp = f"{d}.decoder.layers.{i}"
        lin(lp + ".self_attn.in_proj", hd, 3 * hd)
        lin(lp + ".self_attn.out_proj", hd, hd)
        ln(lp + ".norm1", hd)
        lin(lp + ".cross_attn.sampling_offsets", hd, nh * 3 * npts * 2)
        lin(lp + ".cross_attn.attention_weights", hd, nh * 3 * npts)
        lin(lp + ".cross_attn.value_proj", hd, hd)
        lin(lp + ".cross_attn.output_proj", hd, hd)
        ln(lp + ".norm2", hd)
        lin(lp + ".linear1", hd, d_ffn)
        lin(lp + ".linear2", d_ffn, hd)
        ln(lp + ".norm3", hd)
        lin(f"{d}.dec_score_head.{i}", hd, nc)
        for j, (a, b) in enumerate(((hd, hd), (hd, hd), (hd, 4))):
            lin(f"{d}.dec_bbox_head.{i}.layers.{j}", a, b)
    lin(f"{d}.query_pos_head.layers.0", 4, 2 * hd)
    lin(f"{d}.query_pos_head.layers.1", 2 * hd, hd)
    lin(f"{d}.enc_output.0", hd, hd)
    ln(f"{d}.enc_output.1", hd)
    lin(f"{d}.enc_score_head", hd, nc)
    for j, (a, b) in enumerate(((hd, hd), (hd, hd), (hd, 4))):
        lin(f"{d}.enc_bbox_head.layers.{j}", a, b)
    return specs


def synthetic_rtdetr(seed: int = 0, nc: int = 80, width: float = 1.0, hd: int = 256, ndl: int = 6, nh: int = 8, npts: int = 4, nq: int = 300,
                     d_ffn: int = 1024, score_bias: float = -1.5, box_scale: float = 0.3) -> dict[str, np.ndarray]:
    """Seeded random fused weights of the RT-DETR-l architecture (no checkpoint is reachable here). Conv / linear weights
    ~ N(0, g^2 / fan_in) with g = sqrt(2) in front of a ReLU, 1.7 in front of a SiLU, 1 otherwise; biases ~ N(0, 0.05^2);
    LayerNorm weights 1 + N(0, 0.1^2). The box heads' last layers are damped (box_scale) so that refined boxes stay near their
    anchors; the score heads' biases are shifted (score_bias) so that a minority of the queries clears conf = 0.25."""
    rng = np.random.default_rng(seed)
    t: dict[str, np.ndarray] = {}
    for name, shape, kind in rtdetr_layer_specs(nc, width, hd, ndl, nh, npts, d_ffn):
        if kind == "ln":
            t[name + ".weight"] = (1 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
            t[name + ".bias"] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
            continue
        fan_in = int(np.prod(shape[1:]))
        g = {"relu": np.sqrt(2.0), "silu": 1.7, "lin": 1.0}[kind]
        w = rng.standard_normal(shape) * (g / np.sqrt(fan_in))
        b = rng.standard_normal(shape[0]) * 0.05
        if name.endswith("bbox_head.layers.2"):
            w *= box_scale
        if "score_head" in name:
            b += score_bias
        if name.endswith("in_proj"):                      # nn.MultiheadAttention keeps in_proj_weight / in_proj_bias as parameters
            t[name + "_weight"], t[name + "_bias"] = w.astype(np.float32), b.astype(np.float32)
        else:
            t[name + ".weight"], t[name + ".bias"] = w.astype(np.float32), b.astype(np.float32)
    t["rtdetr.meta"] = np.asarray([nh, npts, nq, 8], np.float32)
    return t


def calibrate_rtdetr_scores(tensors: dict[str, np.ndarray], raw_logits: np.ndarray, conf: float, target: int) -> dict[str, np.ndarray]:
    """A copy of `tensors` whose last decoder score head is shifted by one constant so that about `target` queries of the probed
    frame clear `conf` (the classes keep their order). raw_logits: [queries, nc] class logits of one pass with the unshifted
    weights (Detector.raw_output(logits=True)[:, 4:])."""
    logit = np.sort(raw_logits.max(1).astype(np.float64))[::-1]
    k = min(max(int(target), 1), len(logit) - 1)
    delta = np.log(conf / (1 - conf)) - 0.5 * (logit[k - 1] + logit[k])
    last = max(int(n.split(".")[3]) for n in tensors if n.startswith("model.28.dec_score_head.") and n.endswith(".bias"))
    out = dict(tensors)
    name = f"model.28.dec_score_head.{last}.bias"
    out[name] = (tensors[name] + np.float32(delta)).astype(np.float32)
    return out
