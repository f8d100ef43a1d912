"""ORACLE -- test infrastructure, not product code.

CPU restatement (torch fp32) of the RT-DETR branch of the reference's detector: what
`model.track(frame, **cfg)` computes before the tracker runs when the model's yaml names RT-DETR
(reference call sites: geotrax/extract.py:222-225 swaps `YOLO` for `RTDETR`, extract.py:153 calls it).

The arithmetic lives in ultralytics>=8.4.80,<9.0 (pyproject.toml:56), which is NOT vendored in /root/reference and is
not installed in the build container. Each function restates, from memory of the public source, the published algorithm of
the named upstream piece, wired as ultralytics' cfg/models/rt-detr/rtdetr-l.yaml:

    stretch()               RTDETRPredictor.pre_transform: LetterBox(imgsz, auto=False, scale_fill=True) = cv2.resize to a square
    RtDetrRef.forward()     nn.modules {HGStem, HGBlock (+LightConv, DWConv), Conv, AIFI (TransformerEncoderLayer, post-norm, GELU,
                            2-D sin-cos embedding), RepC3 / RepConv (fused), RTDETRDecoder (input_proj, anchors, enc_output, top-k
                            query selection, DeformableTransformerDecoder with MSDeformAttn = multi_scale_deformable_attn_pytorch,
                            iterative box refinement, eval_idx = -1)}
    postprocess()           RTDETRPredictor.postprocess: xywh -> xyxy, max class score > conf, class filter, descending score,
                            scale by the ORIGINAL frame's width / height (boxes are normalised)

PARITY UNPINNED at this boundary: the reference's own tests never run a model, no RT-DETR weights and no
clip are in the tree, ultralytics cannot be imported here. Pinned only indirectly: conv / pool / layer-norm / attention /
grid_sample are torch's own operators. The stated choices where memory of the source could be wrong: BatchNorm eps 1e-3 everywhere
(ultralytics' initialize_weights), the AIFI embedding built with meshgrid(w, h, indexing="ij") exactly as upstream (which
transposes it against the token order on non-square maps), `valid_mask * feats` before enc_output, top-k on the max class logit,
no NMS, RepConv evaluated in its fused (deploy) form like AutoBackend's model.fuse().

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .yolov8_ref import resize_linear_u8


def stretch(frame_bgr: np.ndarray, imgsz: int) -> torch.Tensor:
    """Frame (HxWx3 BGR u8) -> [1,3,imgsz,imgsz] RGB /255: cv2.resize(INTER_LINEAR) to the square, no padding."""
    img = resize_linear_u8(frame_bgr, imgsz, imgsz)
    rgb = np.ascontiguousarray(img[..., ::-1].transpose(2, 0, 1))
    return (torch.from_numpy(rgb).float() / 255)[None]


def sincos_2d(w: int, h: int, dim: int, temperature: float = 10000.0) -> torch.Tensor:
    """AIFI.build_2d_sincos_position_embedding(w, h, embed_dim) -> [w*h, dim]."""
    gw, gh = torch.meshgrid(torch.arange(w, dtype=torch.float32), torch.arange(h, dtype=torch.float32), indexing="ij")
    pd = dim // 4
    omega = 1.0 / (temperature ** (torch.arange(pd, dtype=torch.float32) / pd))
    ow = gw.flatten()[:, None] @ omega[None]
    oh = gh.flatten()[:, None] @ omega[None]
    return torch.cat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], 1)


def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def ms_deform_attn_core(value, shapes, loc, weights):
    """multi_scale_deformable_attn_pytorch. value [B, S, nh, hd]; loc [B, Q, nh, L, P, 2] in [0,1]; weights [B, Q, nh, L, P]."""
    B, _, nh, hd = value.shape
    _, Q, _, L, P, _ = loc.shape
    vals = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * loc - 1
    sampled = []
    for l, (h, w) in enumerate(shapes):
        v = vals[l].flatten(2).transpose(1, 2).reshape(B * nh, hd, h, w)
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)          # [B*nh, Q, P, 2]
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = weights.transpose(1, 2).reshape(B * nh, 1, Q, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(B, nh * hd, Q)
    return out.transpose(1, 2).contiguous()


class RtDetrRef:
    """RT-DETR (rtdetr-l topology; widths / class count read off the tensors) from a flat dict of fused tensors (ultralytics
    state_dict names; Conv+BN folded, RepConv fused to `.conv`, input_proj folded to `.0.weight/.0.bias`: weights.load_weights)."""

    def __init__(self, tensors: dict[str, np.ndarray]):
        self.t = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in tensors.items()}
        meta = tensors.get("rtdetr.meta")
        self.nh, self.npts, self.nq, self.enc_heads = (int(v) for v in (meta if meta is not None else (8, 4, 300, 8)))
        self.nc = int(self.t["model.28.enc_score_head.weight"].shape[0])
        self.hd = int(self.t["model.28.enc_score_head.weight"].shape[1])
        self.ndl = 0
        while f"model.28.decoder.layers.{self.ndl}.linear1.weight" in self.t:
            self.ndl += 1
        self.acts: dict[str, torch.Tensor] = {}

    # ---- building blocks
    def _conv(self, name, x, stride=1, act="silu", pad=None):
        w, b = self.t[name + ".weight"], self.t.get(name + ".bias")
        k = w.shape[-1]
        y = F.conv2d(x, w, b, stride=stride, padding=k // 2 if pad is None else pad, groups=x.shape[1] // w.shape[1])
        if act == "silu":
            y = F.silu(y)
        elif act == "relu":
            y = F.relu(y)
        self.acts[name] = y
        return y

    def _lin(self, name, x):
        return F.linear(x, self.t[name + ".weight"], self.t.get(name + ".bias"))

    def _ln(self, name, x):
        return F.layer_norm(x, (x.shape[-1],), self.t[name + ".weight"], self.t[name + ".bias"], 1e-5)

    def _mlp(self, name, x, n):
        for i in range(n):
            x = self._lin(f"{name}.layers.{i}", x)
            if i < n - 1:
                x = F.relu(x)
        return x

    def _mha(self, name, q, k, v, nh):
        """nn.MultiheadAttention (batch-first view): q, k, v [B, T, C]."""
        C = q.shape[-1]
        W, bias = self.t[name + ".in_proj_weight"], self.t[name + ".in_proj_bias"]
        qp = F.linear(q, W[:C], bias[:C])
        kp = F.linear(k, W[C:2 * C], bias[C:2 * C])
        vp = F.linear(v, W[2 * C:], bias[2 * C:])
        B, T, _ = qp.shape
        S = kp.shape[1]
        d = C // nh
        qh = qp.view(B, T, nh, d).transpose(1, 2) / math.sqrt(d)     # torch scales q before the product
        kh = kp.view(B, S, nh, d).transpose(1, 2)
        vh = vp.view(B, S, nh, d).transpose(1, 2)
        a = (qh @ kh.transpose(-1, -2)).softmax(-1)
        o = (a @ vh).transpose(1, 2).reshape(B, T, C)
        return self._lin(name + ".out_proj", o)

    def _hgstem(self, p, x):
        x = self._conv(p + ".stem1.conv", x, 2, "relu")
        x = F.pad(x, [0, 1, 0, 1])
        x2 = self._conv(p + ".stem2a.conv", x, 1, "relu", pad=0)
        x2 = F.pad(x2, [0, 1, 0, 1])
        x2 = self._conv(p + ".stem2b.conv", x2, 1, "relu", pad=0)
        x1 = F.max_pool2d(x, kernel_size=2, stride=1, padding=0, ceil_mode=True)
        x = torch.cat([x1, x2], 1)
        x = self._conv(p + ".stem3.conv", x, 2, "relu")
        return self._conv(p + ".stem4.conv", x, 1, "relu")

    def _hgblock(self, p, x, shortcut):
        y = [x]
        i = 0
        while True:
            if f"{p}.m.{i}.conv.weight" in self.t:                    # Conv(k) + ReLU
                y.append(self._conv(f"{p}.m.{i}.conv", y[-1], 1, "relu"))
            elif f"{p}.m.{i}.conv1.conv.weight" in self.t:            # LightConv: 1x1 (no act) then depthwise k x k + ReLU
                h = self._conv(f"{p}.m.{i}.conv1.conv", y[-1], 1, None)
                y.append(self._conv(f"{p}.m.{i}.conv2.conv", h, 1, "relu"))
            else:
                break
            i += 1
        out = self._conv(p + ".ec.conv", self._conv(p + ".sc.conv", torch.cat(y, 1), 1, "relu"), 1, "relu")
        out = out + x if shortcut and x.shape[1] == out.shape[1] else out
        self.acts[p] = out
        return out

    def _aifi(self, p, x):
        B, C, H, W = x.shape
        pos = sincos_2d(W, H, C)[None]
        src = x.flatten(2).permute(0, 2, 1)
        q = src + pos
        src = self._ln(p + ".norm1", src + self._mha(p + ".ma", q, q, src, self.enc_heads))
        ff = self._lin(p + ".fc2", F.gelu(self._lin(p + ".fc1", src)))
        src = self._ln(p + ".norm2", src + ff)
        out = src.permute(0, 2, 1).reshape(B, C, H, W).contiguous()
        self.acts[p] = out
        return out

    def _repc3(self, p, x):
        a = self._conv(p + ".cv1.conv", x)
        i = 0
        while f"{p}.m.{i}.conv.weight" in self.t:
            a = self._conv(f"{p}.m.{i}.conv", a)                      # fused RepConv: 3x3 (+1x1 at the centre) + SiLU
            i += 1
        out = a + self._conv(p + ".cv2.conv", x)
        if p + ".cv3.conv.weight" in self.t:
            out = self._conv(p + ".cv3.conv", out)
        self.acts[p] = out
        return out

    # ---- decoder
    def _anchors(self, shapes, grid_size=0.05, eps=1e-2):
        anchors = []
        for i, (h, w) in enumerate(shapes):
            gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
            xy = (torch.stack([gx, gy], -1)[None] + 0.5) / torch.tensor([w, h], dtype=torch.float32)
            wh = torch.ones_like(xy) * grid_size * (2.0 ** i)
            anchors.append(torch.cat([xy, wh], -1).view(-1, h * w, 4))
        anchors = torch.cat(anchors, 1)
        valid = ((anchors > eps) & (anchors < 1 - eps)).all(-1, keepdim=True)
        anchors = torch.log(anchors / (1 - anchors)).masked_fill(~valid, float("inf"))
        return anchors, valid

    def _decoder(self, feats3):
        p = "model.28"
        proj = [F.conv2d(f, self.t[f"{p}.input_proj.{i}.0.weight"], self.t[f"{p}.input_proj.{i}.0.bias"]) for i, f in enumerate(feats3)]
        shapes = [tuple(f.shape[2:]) for f in proj]
        feats = torch.cat([f.flatten(2).permute(0, 2, 1) for f in proj], 1)          # [B, S, hd]
        self.acts[p + ".feats"] = feats
        B = feats.shape[0]
        anchors, valid = self._anchors(shapes)
        enc = self._ln(p + ".enc_output.1", self._lin(p + ".enc_output.0", valid * feats))
        self.acts[p + ".enc_output"] = enc
        scores = self._lin(p + ".enc_score_head", enc)
        self.acts[p + ".enc_scores"] = scores
        topk = torch.topk(scores.max(-1).values, self.nq, dim=1).indices            # [B, nq]
        self.topk = topk
        bi = torch.arange(B)[:, None]
        embed = enc[bi, topk]
        refer = self._mlp(p + ".enc_bbox_head", embed, 3) + anchors[0][topk]
        self.acts[p + ".refer0"] = refer
        refer = refer.sigmoid()
        out = embed
        for i in range(self.ndl):
            lp = f"{p}.decoder.layers.{i}"
            qpos = self._mlp(p + ".query_pos_head", refer, 2)
            q = out + qpos
            out = self._ln(lp + ".norm1", out + self._mha(lp + ".self_attn", q, q, out, self.nh))
            # cross attention: multi-scale deformable sampling around the reference boxes
            query = out + qpos
            value = self._lin(lp + ".cross_attn.value_proj", feats).view(B, -1, self.nh, self.hd // self.nh)
            L, P = len(shapes), self.npts
            off = self._lin(lp + ".cross_attn.sampling_offsets", query).view(B, self.nq, self.nh, L, P, 2)
            aw = self._lin(lp + ".cross_attn.attention_weights", query).view(B, self.nq, self.nh, L * P).softmax(-1).view(B, self.nq, self.nh, L, P)
            rb = refer[:, :, None, None, None, :]
            loc = rb[..., :2] + off / P * rb[..., 2:] * 0.5
            ca = self._lin(lp + ".cross_attn.output_proj", ms_deform_attn_core(value, shapes, loc, aw))
            out = self._ln(lp + ".norm2", out + ca)
            out = self._ln(lp + ".norm3", out + self._lin(lp + ".linear2", F.relu(self._lin(lp + ".linear1", out))))
            self.acts[lp] = out
            refer = torch.sigmoid(self._mlp(f"{p}.dec_bbox_head.{i}", out, 3) + inverse_sigmoid(refer))
            self.acts[lp + ".refer"] = refer
        cls = self._lin(f"{p}.dec_score_head.{self.ndl - 1}", out)
        return torch.cat([refer, cls.sigmoid()], -1)                                  # [B, nq, 4 + nc]: xywh normalised + scores

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        a = self.acts
        x0 = self._hgstem("model.0", x)
        a["model.0"] = x0
        x1 = self._hgblock("model.1", x0, False)
        x2 = self._conv("model.2.conv", x1, 2, None)
        x3 = self._hgblock("model.3", x2, False)
        x4 = self._conv("model.4.conv", x3, 2, None)
        x5 = self._hgblock("model.5", x4, False)
        x6 = self._hgblock("model.6", x5, True)
        x7 = self._hgblock("model.7", x6, True)
        x8 = self._conv("model.8.conv", x7, 2, None)
        x9 = self._hgblock("model.9", x8, False)
        x10 = self._conv("model.10.conv", x9, 1, None)
        x11 = self._aifi("model.11", x10)
        x12 = self._conv("model.12.conv", x11)
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
        x14 = self._conv("model.14.conv", x7, 1, None)
        x16 = self._repc3("model.16", torch.cat([up(x12), x14], 1))
        x17 = self._conv("model.17.conv", x16)
        x19 = self._conv("model.19.conv", x3, 1, None)
        x21 = self._repc3("model.21", torch.cat([up(x17), x19], 1))
        x22 = self._conv("model.22.conv", x21, 2)
        x24 = self._repc3("model.24", torch.cat([x22, x17], 1))
        x25 = self._conv("model.25.conv", x24, 2)
        x27 = self._repc3("model.27", torch.cat([x25, x12], 1))
        return self._decoder((x21, x24, x27))


def postprocess(pred: np.ndarray, frame_hw, conf: float, classes=None, max_det: int = 300):
    """pred [nq, 4 + nc] of one image -> xyxy [n,4] frame pixels, conf [n], cls [n], query index [n] (descending score, stable)."""
    pred = pred.astype(np.float32)
    xy, wh = pred[:, :2], pred[:, 2:4]
    box = np.concatenate([xy - wh / np.float32(2), xy + wh / np.float32(2)], 1)
    score = pred[:, 4:].max(1)
    cls = pred[:, 4:].argmax(1)
    keep = score > np.float32(conf)
    if classes is not None:
        keep &= np.isin(cls, np.asarray(classes))
    idx = np.flatnonzero(keep)
    idx = idx[np.argsort(-score[idx], kind="stable")][:max_det]
    h, w = frame_hw
    b = box[idx].copy()
    b[:, [0, 2]] *= np.float32(w)
    b[:, [1, 3]] *= np.float32(h)
    return b, score[idx], cls[idx].astype(np.int32), idx


def detect(model: RtDetrRef, frame_bgr: np.ndarray, imgsz: int, conf: float, classes=None, max_det: int = 300):
    pred = model.forward(stretch(frame_bgr, imgsz))[0].numpy()
    return postprocess(pred, frame_bgr.shape[:2], conf, classes, max_det)[:3]
