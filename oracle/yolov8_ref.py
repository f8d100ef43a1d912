"""ORACLE -- test infrastructure, not product code.

CPU restatement (torch fp32 + numpy) of the detector half of the reference's hot path:
what `model.track(frame, **cfg)` computes before the tracker runs
(reference call site: geotrax/extract.py:153; model loaded at extract.py:222).

The arithmetic lives in third-party packages that are NOT vendored in /root/reference and are
not installed in the build container: ultralytics>=8.4.80,<9.0 (pyproject.toml:56), torchvision
(NMS) and OpenCV (resize / cvtColor / copyMakeBorder). Each function below restates the
published algorithm of the named upstream function from memory of its public source:

    letterbox()            ultralytics.data.augment.LetterBox.__call__ + predictor.preprocess
    YoloV8Ref.forward()    ultralytics.nn.modules {Conv.forward_fuse, C2f, Bottleneck, SPPF,
                           Detect (+DFL, make_anchors, dist2bbox)} wired as cfg/models/v8/yolov8.yaml
    non_max_suppression()  ultralytics.utils.ops.non_max_suppression + torchvision.ops.nms
    scale_boxes()          ultralytics.utils.ops.scale_boxes / clip_boxes
    bgr2gray_half()        cv2.cvtColor(BGR2GRAY) + cv2.resize(0.5) as stabilo applies them

PARITY UNPINNED at this boundary: the reference's own tests never run a model,
the weights and the clip are absent, and neither ultralytics nor cv2 can be imported here, so
nothing in this file could be checked against the real packages. It is pinned only indirectly:
conv/pool/upsample are checked against torch.nn.functional itself, and the end-to-end
post-processing chain is pinned by the reference's golden outputs (tests/golden/). The general (non-2x) bilinear of the
letterbox is held against skimage.transform.resize(order=1, anti_aliasing=False): <= 1 grey level
(tests/test_independent.py; scikit-image is not a dependency of the reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- preprocessing

def _py_round(v: float) -> int:
    return int(round(v))  # Python round: ties to even, like the upstream code


def letterbox_geometry(src_h: int, src_w: int, imgsz: int, rect: bool, stride: int = 32):
    """LetterBox(new_shape=imgsz, auto=rect, center=True, scaleup=True) geometry."""
    r = min(imgsz / src_h, imgsz / src_w)
    new_w, new_h = _py_round(src_w * r), _py_round(src_h * r)
    dw, dh = imgsz - new_w, imgsz - new_h
    if rect:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = _py_round(dh - 0.1), _py_round(dh + 0.1)
    left, right = _py_round(dw - 0.1), _py_round(dw + 0.1)
    return dict(new_h=new_h, new_w=new_w, top=top, bottom=bottom, left=left, right=right,
                net_h=new_h + top + bottom, net_w=new_w + left + right)


def resize_linear_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC.

    Exact 2x reduction: OpenCV's INTER_LINEAR switches to its 2x2 area kernel,
    (a+b+c+d+2)>>2. Otherwise: 11-bit fixed-point bilinear (resize.cpp HResizeLinear /
    VResizeLinear) [restated from memory, unverified]."""
    h, w, _ = img.shape
    if (new_h, new_w) == (h, w):
        return img.copy()
    if new_h * 2 == h and new_w * 2 == w:
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, sy = np.float32(w / new_w), np.float32(h / new_h)

    def coeffs(n_dst, n_src, scale):
        f = (np.arange(n_dst, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
        i0 = np.floor(f).astype(np.int64)
        f = f - i0.astype(np.float32)
        lo, hi = i0 < 0, i0 >= n_src - 1
        f[lo | hi] = 0
        i0[lo] = 0
        i0[hi] = n_src - 1
        i1 = np.minimum(i0 + 1, n_src - 1)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return i0, i1, a0, a1

    x0, x1, ax0, ax1 = coeffs(new_w, w, sx)
    y0, y1, ay0, ay1 = coeffs(new_h, h, sy)
    s = img.astype(np.int64)
    rows = s[:, x0, :] * ax0[None, :, None] + s[:, x1, :] * ax1[None, :, None]
    r0, r1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(frame_bgr: np.ndarray, imgsz: int, rect: bool, half: bool = False) -> tuple[torch.Tensor, dict]:
    """Frame (HxWx3 BGR u8) -> network input tensor [1,3,net_h,net_w] (RGB, /255)."""
    h, w, _ = frame_bgr.shape
    g = letterbox_geometry(h, w, imgsz, rect)
    img = resize_linear_u8(frame_bgr, g["new_h"], g["new_w"])
    canvas = np.full((g["net_h"], g["net_w"], 3), 114, dtype=np.uint8)
    canvas[g["top"]:g["top"] + g["new_h"], g["left"]:g["left"] + g["new_w"]] = img
    rgb = np.ascontiguousarray(canvas[..., ::-1].transpose(2, 0, 1))
    t = torch.from_numpy(rgb)
    if half:
        t = (t.half() / 255).float()  # ultralytics: im.half(); im /= 255
    else:
        t = t.float() / 255
    return t[None], g


def bgr2gray_half(frame_bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(BGR2GRAY) (14-bit fixed point) followed by the exact 2x reduction."""
    f = frame_bgr.astype(np.int32)
    g = (f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14
    return ((g[0::2, 0::2] + g[0::2, 1::2] + g[1::2, 0::2] + g[1::2, 1::2] + 2) >> 2).astype(np.uint8)


# --------------------------------------------------------------------------- network

class YoloV8Ref:
    """YOLOv8 detect model evaluated from a flat dict of *fused* tensors (ultralytics state_dict
    names after Conv/BN fusion). `emulate_half` rounds weights and every layer output to fp16,
    which is where the fp16 HIP path rounds (accumulation stays fp32 in both)."""

    def __init__(self, tensors: dict[str, np.ndarray], emulate_half: bool = False):
        self.t = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in tensors.items()}
        self.half = emulate_half
        self.nc = int(self.t["model.22.cv3.0.2.weight"].shape[0])
        self.acts: dict[str, torch.Tensor] = {}

    def _q(self, x: torch.Tensor) -> torch.Tensor:
        return x.half().float() if self.half else x

    def _conv(self, name: str, x: torch.Tensor, stride: int = 1, act: bool = True, quant_out: bool = True) -> torch.Tensor:
        w = self._q(self.t[name + ".weight"])
        b = self.t.get(name + ".bias")
        y = F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)
        if act:
            y = F.silu(y)
        return self._q(y) if quant_out else y

    def _conv_res(self, name: str, x: torch.Tensor, res: torch.Tensor | None) -> torch.Tensor:
        # Bottleneck tail: x + cv2(cv1(x)); the HIP kernel adds the residual in fp32 before the
        # single rounding of the stored value.
        y = self._conv(name, x, quant_out=False)
        if res is not None:
            y = y + res
        return self._q(y)

    def _c2f(self, pfx: str, x: torch.Tensor, shortcut: bool) -> torch.Tensor:
        y = list(self._conv(pfx + ".cv1.conv", x).chunk(2, 1))
        k = 0
        while f"{pfx}.m.{k}.cv1.conv.weight" in self.t:
            inp = y[-1]
            h = self._conv(f"{pfx}.m.{k}.cv1.conv", inp)
            self.acts[f"{pfx}.m.{k}.cv1.conv"] = h
            y.append(self._conv_res(f"{pfx}.m.{k}.cv2.conv", h, inp if shortcut else None))
            k += 1
        out = self._conv(pfx + ".cv2.conv", torch.cat(y, 1))
        self.acts[pfx] = out
        return out

    def _sppf(self, pfx: str, x: torch.Tensor) -> torch.Tensor:
        y = [self._conv(pfx + ".cv1.conv", x)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        self.acts[pfx + ".pools"] = torch.cat(y, 1)
        out = self._conv(pfx + ".cv2.conv", torch.cat(y, 1))
        self.acts[pfx] = out
        return out

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B,3,H,W] float -> [B, A, 4+nc]: xywh in network pixels + sigmoid class scores
        (the transpose of ultralytics' Detect inference output)."""
        a = self.acts
        x = self._q(x)
        a["model.0.conv"] = x0 = self._conv("model.0.conv", x, 2)
        a["model.1.conv"] = x1 = self._conv("model.1.conv", x0, 2)
        x2 = self._c2f("model.2", x1, True)
        a["model.3.conv"] = x3 = self._conv("model.3.conv", x2, 2)
        x4 = self._c2f("model.4", x3, True)
        a["model.5.conv"] = x5 = self._conv("model.5.conv", x4, 2)
        x6 = self._c2f("model.6", x5, True)
        a["model.7.conv"] = x7 = self._conv("model.7.conv", x6, 2)
        x8 = self._c2f("model.8", x7, True)
        x9 = self._sppf("model.9", x8)
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
        x12 = self._c2f("model.12", torch.cat([up(x9), x6], 1), False)
        x15 = self._c2f("model.15", torch.cat([up(x12), x4], 1), False)
        a["model.16.conv"] = x16 = self._conv("model.16.conv", x15, 2)
        x18 = self._c2f("model.18", torch.cat([x16, x12], 1), False)
        a["model.19.conv"] = x19 = self._conv("model.19.conv", x18, 2)
        x21 = self._c2f("model.21", torch.cat([x19, x9], 1), False)

        self.detect_inputs = (x15, x18, x21)        # what ultralytics' `with_reid, model: auto` hook keeps (the Detect layer's inputs)
        outs = []
        for l, (f, stride) in enumerate(zip((x15, x18, x21), (8.0, 16.0, 32.0))):
            b = self._conv(f"model.22.cv2.{l}.1.conv", self._conv(f"model.22.cv2.{l}.0.conv", f))
            c = self._conv(f"model.22.cv3.{l}.1.conv", self._conv(f"model.22.cv3.{l}.0.conv", f))
            a[f"model.22.feat{l}"] = torch.cat([b, c], 1)
            # final 1x1 convs run in fp32 on the stored features (the HIP decode kernels do too)
            box = F.conv2d(b, self.t[f"model.22.cv2.{l}.2.weight"], self.t[f"model.22.cv2.{l}.2.bias"])
            cls = F.conv2d(c, self.t[f"model.22.cv3.{l}.2.weight"], self.t[f"model.22.cv3.{l}.2.bias"])
            B, _, H, W = box.shape
            # DFL: softmax over 16 bins per side, expectation with weights arange(16)
            p = box.view(B, 4, 16, H * W).softmax(2)
            d = (p * torch.arange(16, dtype=torch.float32).view(1, 1, 16, 1)).sum(2)  # [B,4,A]
            ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32) + 0.5,
                                    torch.arange(W, dtype=torch.float32) + 0.5, indexing="ij")
            anc = torch.stack([xs.reshape(-1), ys.reshape(-1)], 0)[None]  # [1,2,A]
            x1y1, x2y2 = anc - d[:, :2], anc + d[:, 2:]
            xywh = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], 1) * stride
            outs.append(torch.cat([xywh, cls.view(B, self.nc, H * W).sigmoid()], 1))
        return torch.cat(outs, 2).transpose(1, 2).contiguous()


    def obj_feats_table(self) -> np.ndarray:
        """ultralytics engine/predictor.py get_obj_feats on the last forward: [A, s] with s = the narrowest level's channel count --
        every level's channels averaged in consecutive groups of C / s, levels concatenated in anchor order."""
        maps = self.detect_inputs
        s = min(int(x.shape[1]) for x in maps)
        return torch.cat([x.permute(0, 2, 3, 1).reshape(x.shape[0], -1, s, x.shape[1] // s).float().mean(dim=-1) for x in maps], dim=1)[0].numpy()


# --------------------------------------------------------------------------- post-processing

def nms_torchvision(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """torchvision.ops.nms (CPU kernel): stable descending sort, suppress IoU > thr. fp32."""
    boxes = boxes.astype(np.float32)
    order = np.argsort(-scores.astype(np.float32), kind="stable")
    x1, y1, x2, y2 = (boxes[:, i] for i in range(4))
    areas = (x2 - x1) * (y2 - y1)
    n = len(order)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1, yy1 = np.maximum(x1[i], x1[rest]), np.maximum(y1[i], y1[rest])
        xx2, yy2 = np.minimum(x2[i], x2[rest]), np.minimum(y2[i], y2[rest])
        inter = np.maximum(np.float32(0), xx2 - xx1) * np.maximum(np.float32(0), yy2 - yy1)
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > np.float32(thr)]] = True
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(pred: np.ndarray, conf: float, iou: float, classes=None, agnostic: bool = False,
                        max_det: int = 300, max_nms: int = 30000, max_wh: int = 7680, return_idx: bool = False):
    """pred: [A, 4+nc] (xywh + scores) of one image -> [n, 6] xyxy, conf, cls (network pixels); return_idx: also the anchor
    index of every kept row (ultralytics non_max_suppression(return_idxs=True))."""
    pred = pred.astype(np.float32)
    xy, wh = pred[:, :2], pred[:, 2:4] / np.float32(2)
    box = np.concatenate([xy - wh, xy + wh], 1)
    cls = pred[:, 4:]
    keep0 = cls.max(1) > np.float32(conf)
    anchors = np.flatnonzero(keep0)
    box, cls = box[keep0], cls[keep0]
    j = cls.argmax(1)
    cf = cls[np.arange(len(j)), j]
    sel = cf > np.float32(conf)
    x, anchors = np.concatenate([box, cf[:, None], j[:, None].astype(np.float32)], 1)[sel], anchors[sel]
    if classes is not None:
        sel = np.isin(x[:, 5].astype(int), np.asarray(classes))
        x, anchors = x[sel], anchors[sel]
    if len(x) == 0:
        return (np.zeros((0, 6), np.float32), np.zeros(0, np.int64)) if return_idx else np.zeros((0, 6), np.float32)
    if len(x) > max_nms:
        sel = np.argsort(-x[:, 4], kind="stable")[:max_nms]
        x, anchors = x[sel], anchors[sel]
    c = x[:, 5:6] * np.float32(0 if agnostic else max_wh)
    i = nms_torchvision(x[:, :4] + c, x[:, 4], iou)[:max_det]
    return (x[i], anchors[i]) if return_idx else x[i]


def scale_boxes(boxes_xyxy: np.ndarray, net_hw: tuple[int, int], src_hw: tuple[int, int]) -> np.ndarray:
    """ultralytics scale_boxes(img1_shape=net, boxes, img0_shape=src) + clip_boxes, fp32."""
    b = boxes_xyxy.astype(np.float32).copy()
    gain = min(net_hw[0] / src_hw[0], net_hw[1] / src_hw[1])
    padx = _py_round((net_hw[1] - src_hw[1] * gain) / 2 - 0.1)
    pady = _py_round((net_hw[0] - src_hw[0] * gain) / 2 - 0.1)
    b[:, [0, 2]] -= np.float32(padx)
    b[:, [1, 3]] -= np.float32(pady)
    b[:, :4] /= np.float32(gain)
    b[:, [0, 2]] = b[:, [0, 2]].clip(0, src_hw[1])
    b[:, [1, 3]] = b[:, [1, 3]].clip(0, src_hw[0])
    return b


def detect(model: YoloV8Ref, frame_bgr: np.ndarray, imgsz: int, rect: bool, conf: float, iou: float,
           classes=None, agnostic: bool = False, max_det: int = 300, return_feats: bool = False):
    """Whole detector chain on one frame -> (xyxy [n,4] frame pixels, conf [n], cls [n]); return_feats: also the [n, s]
    appearance vectors BoT-SORT's `model: auto` ReID reads (obj_feats_table at the kept anchors)."""
    x, g = letterbox(frame_bgr, imgsz, rect, half=model.half)
    pred = model.forward(x)[0].numpy()
    det, idx = non_max_suppression(pred, conf, iou, classes, agnostic, max_det, return_idx=True)
    xyxy = scale_boxes(det[:, :4], (g["net_h"], g["net_w"]), frame_bgr.shape[:2])
    if return_feats:
        return xyxy, det[:, 4], det[:, 5].astype(np.int32), model.obj_feats_table()[idx]
    return xyxy, det[:, 4], det[:, 5].astype(np.int32)


# --------------------------------------------------------------------------- single operators

def conv2d_nhwc(x: np.ndarray, w_ohwi: np.ndarray, bias=None, stride: int = 1, act: bool = True,
                residual=None) -> np.ndarray:
    """ultralytics Conv.forward_fuse on NHWC data, computed in fp32 with torch (the checker for
    gtx_op_conv2d). x may be fp16: it is widened exactly, the result is returned as fp32. float64 inputs are
    computed and returned in float64 (the yardstick for the fp32-grade kernels' own rounding error)."""
    dt = np.float64 if x.dtype == np.float64 else np.float32
    xt = torch.from_numpy(np.ascontiguousarray(x.astype(dt))).permute(0, 3, 1, 2)
    wt = torch.from_numpy(np.ascontiguousarray(w_ohwi.astype(dt))).permute(0, 3, 1, 2)
    bt = None if bias is None else torch.from_numpy(np.asarray(bias, dtype=dt))
    with torch.no_grad():
        y = F.conv2d(xt, wt, bt, stride=stride, padding=w_ohwi.shape[1] // 2)
        if act:
            y = F.silu(y)
        y = y.permute(0, 2, 3, 1)
        if residual is not None:
            y = y + torch.from_numpy(residual.astype(dt))
    return y.contiguous().numpy()


def sppf_pools_nhwc(x: np.ndarray) -> np.ndarray:
    """[n,h,w,c] -> [n,h,w,4c]: x and its three cascaded 5x5/s1/p2 max-pools (SPPF.forward)."""
    xt = torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2)
    ys = [xt]
    for _ in range(3):
        ys.append(F.max_pool2d(ys[-1], 5, 1, 2))
    return torch.cat(ys, 1).permute(0, 2, 3, 1).contiguous().numpy()


def upsample2x_nhwc(x: np.ndarray) -> np.ndarray:
    return x.repeat(2, axis=1).repeat(2, axis=2)
