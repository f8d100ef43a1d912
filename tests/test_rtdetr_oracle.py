"""CPU checks around the RT-DETR oracle (oracle/rtdetr_ref.py) and the weight plumbing in front of it: the pieces that can be held
against torch's own operators or a brute-force restatement are (the oracle as a whole is PARITY UNPINNED: ultralytics is absent)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def test_fused_repconv_and_folded_projection_equal_the_unfused_modules():
    """weights.fuse_repconv / fold_input_proj (what load_weights does to a checkpoint) against conv3x3 + conv1x1 and Conv2d + BatchNorm2d(eval)."""
    from geotrax_amd.weights import BN_EPS, fold_bn, fold_input_proj, fuse_repconv, is_rtdetr

    rng = np.random.default_rng(0)
    c = 16
    t = {"model.16.m.0.conv1.conv.weight": rng.standard_normal((c, c, 3, 3)).astype(np.float32), "model.16.m.0.conv1.conv.bias": rng.standard_normal(c).astype(np.float32),
         "model.16.m.0.conv2.conv.weight": rng.standard_normal((c, c, 1, 1)).astype(np.float32), "model.16.m.0.conv2.conv.bias": rng.standard_normal(c).astype(np.float32),
         # LightConv has the same tensor names with other shapes (1x1 then depthwise): must be left alone
         "model.5.m.0.conv1.conv.weight": rng.standard_normal((c, c, 1, 1)).astype(np.float32), "model.5.m.0.conv2.conv.weight": rng.standard_normal((c, 1, 5, 5)).astype(np.float32),
         "model.28.input_proj.0.0.weight": rng.standard_normal((c, c, 1, 1)).astype(np.float32), "model.28.input_proj.0.1.weight": rng.uniform(0.5, 1.5, c).astype(np.float32),
         "model.28.input_proj.0.1.bias": rng.standard_normal(c).astype(np.float32), "model.28.input_proj.0.1.running_mean": rng.standard_normal(c).astype(np.float32),
         "model.28.input_proj.0.1.running_var": rng.uniform(0.5, 2.0, c).astype(np.float32), "model.28.decoder.layers.0.linear1.weight": np.zeros((4, c), np.float32)}
    assert is_rtdetr(t)
    f = fold_input_proj(fuse_repconv(fold_bn(t)))
    assert "model.16.m.0.conv.weight" in f and "model.16.m.0.conv1.conv.weight" not in f
    assert "model.5.m.0.conv1.conv.weight" in f and "model.5.m.0.conv2.conv.weight" in f
    x = torch.from_numpy(rng.standard_normal((1, c, 9, 11)).astype(np.float32))
    T = lambda k: torch.from_numpy(t[k])
    want = F.conv2d(x, T("model.16.m.0.conv1.conv.weight"), T("model.16.m.0.conv1.conv.bias"), padding=1) + \
        F.conv2d(x, T("model.16.m.0.conv2.conv.weight"), T("model.16.m.0.conv2.conv.bias"))
    got = F.conv2d(x, torch.from_numpy(f["model.16.m.0.conv.weight"]), torch.from_numpy(f["model.16.m.0.conv.bias"]), padding=1)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-5)
    y = F.conv2d(x, T("model.28.input_proj.0.0.weight"))
    want = F.batch_norm(y, T("model.28.input_proj.0.1.running_mean"), T("model.28.input_proj.0.1.running_var"), T("model.28.input_proj.0.1.weight"),
                        T("model.28.input_proj.0.1.bias"), False, 0.0, BN_EPS)
    got = F.conv2d(x, torch.from_numpy(f["model.28.input_proj.0.0.weight"]), torch.from_numpy(f["model.28.input_proj.0.0.bias"]))
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-5)


def test_deformable_attention_core_equals_a_brute_force_bilinear_sum():
    """oracle.rtdetr_ref.ms_deform_attn_core (grid_sample form, as upstream) against explicit four-tap sampling with zero padding."""
    from oracle.rtdetr_ref import ms_deform_attn_core

    rng = np.random.default_rng(1)
    B, nh, hd, Q, P = 1, 2, 4, 5, 3
    shapes = [(6, 7), (3, 4)]
    S = sum(h * w for h, w in shapes)
    value = torch.from_numpy(rng.standard_normal((B, S, nh, hd)).astype(np.float32))
    loc = torch.from_numpy(rng.uniform(-0.2, 1.2, (B, Q, nh, len(shapes), P, 2)).astype(np.float32))      # some points fall outside: zeros
    w = torch.from_numpy(rng.uniform(0, 1, (B, Q, nh, len(shapes), P)).astype(np.float32))
    got = ms_deform_attn_core(value, shapes, loc, w).numpy()
    want = np.zeros((B, Q, nh * hd), np.float64)
    v = value.numpy().astype(np.float64)
    off = 0
    for l, (h, wd) in enumerate(shapes):
        img = v[0, off:off + h * wd].reshape(h, wd, nh, hd)
        off += h * wd
        for q in range(Q):
            for hh in range(nh):
                for p in range(P):
                    x = float(loc[0, q, hh, l, p, 0]) * wd - 0.5
                    y = float(loc[0, q, hh, l, p, 1]) * h - 0.5
                    x0, y0 = int(np.floor(x)), int(np.floor(y))
                    acc = np.zeros(hd)
                    for yy, xx, wt in ((y0, x0, (1 - (x - x0)) * (1 - (y - y0))), (y0, x0 + 1, (x - x0) * (1 - (y - y0))),
                                       (y0 + 1, x0, (1 - (x - x0)) * (y - y0)), (y0 + 1, x0 + 1, (x - x0) * (y - y0))):
                        if 0 <= yy < h and 0 <= xx < wd:
                            acc += img[yy, xx, hh] * wt
                    want[0, q, hh * hd:(hh + 1) * hd] += acc * float(w[0, q, hh, l, p])
    np.testing.assert_allclose(got, want, atol=1e-5)


def test_oracle_forward_on_a_narrow_model_and_the_score_stage():
    """A quarter-width, two-layer variant (widths read off the tensors) runs through the oracle; postprocess filters, orders and scales."""
    from geotrax_amd.weights import calibrate_rtdetr_scores, synthetic_rtdetr
    from oracle.rtdetr_ref import RtDetrRef, postprocess, sincos_2d, stretch

    t = synthetic_rtdetr(seed=5, nc=3, width=0.25, hd=64, ndl=2, nq=50, d_ffn=128)
    ref = RtDetrRef(t)
    assert (ref.hd, ref.nc, ref.ndl, ref.nq) == (64, 3, 2, 50)
    frame = np.random.default_rng(0).integers(0, 255, (96, 160, 3), dtype=np.uint8)
    pred = ref.forward(stretch(frame, 128))[0].numpy()
    assert pred.shape == (50, 7) and np.isfinite(pred).all() and (pred[:, 4:] > 0).all() and (pred[:, 4:] < 1).all()
    xyxy, score, cls, idx = postprocess(pred, frame.shape[:2], 0.0, classes=[0, 2], max_det=10)
    assert len(score) <= 10 and (np.diff(score) <= 0).all() and set(cls) <= {0, 2}
    k = idx[0]
    cx, cy, w, h = pred[k, :4]
    np.testing.assert_allclose(xyxy[0], [(cx - w / 2) * 160, (cy - h / 2) * 96, (cx + w / 2) * 160, (cy + h / 2) * 96], rtol=1e-6)
    # calibrate_rtdetr_scores: one shift of the last score head, classes keep their order, about `target` queries clear conf
    logits = np.log(pred[:, 4:] / (1 - pred[:, 4:]))
    t2 = calibrate_rtdetr_scores(t, logits, 0.25, 12)
    pred2 = RtDetrRef(t2).forward(stretch(frame, 128))[0].numpy()
    assert abs(int((pred2[:, 4:].max(1) > 0.25).sum()) - 12) <= 1 and (pred2[:, 4:].argmax(1) == pred[:, 4:].argmax(1)).all()
    # AIFI's embedding: the four quarters are sin / cos of (index / h) and (index % h) scaled by the 10000^(k / pd) ladder
    e = sincos_2d(3, 2, 8).numpy()
    assert e.shape == (6, 8)
    np.testing.assert_allclose(e[:, 0], np.sin(np.arange(6) // 2), atol=1e-6)
    np.testing.assert_allclose(e[:, 4], np.sin(np.arange(6) % 2), atol=1e-6)
    np.testing.assert_allclose(e[:, 3], np.cos((np.arange(6) // 2) / 100.0), atol=1e-6)
