#!/usr/bin/env python3
"""Converts an ultralytics YOLOv8 or RT-DETR detect checkpoint (.pt) into the flat .safetensors file +
`<name>.names.yaml` side-car that geotrax_amd.model.YOLO / RTDETR load.

Run this once on a machine where `ultralytics` is installed (it is not part of this build):

    python tools/convert_weights.py geotrax_hbb_yolov8s_1920_v1.pt [out.safetensors]

Conv+BN pairs are fused with ultralytics' own `model.fuse()`; tensor names are the fused model's
state_dict keys (`model.0.conv.weight`, `model.0.conv.bias`, ... `model.22.cv3.2.2.bias`).

An RT-DETR checkpoint (the reference's `RTDETR` branch, geotrax/extract.py:222-225; rtdetr-l topology) keeps its state_dict names
too (`model.0.stem1.conv.weight` ... `model.28.decoder.layers.5.norm3.bias`); what `fuse()` leaves unfused -- RepConv's two
branches where a version does not fuse them, the decoder's Sequential(Conv2d, BatchNorm2d) input projections -- is folded by
geotrax_amd.weights.load_weights when the file is read. The decoder's head / point / query counts, which no tensor shape carries,
go into the small `rtdetr.meta` tensor [heads, points, queries, AIFI heads].
"""
import sys
from pathlib import Path

import yaml


def main():
    from safetensors.torch import save_file
    from ultralytics import YOLO

    src = Path(sys.argv[1])
    dst = Path(sys.argv[2]) if len(sys.argv) > 2 else src.with_suffix(".safetensors")
    yolo = YOLO(str(src))
    if "rtdetr" in str(getattr(yolo.model, "yaml_file", "") or getattr(yolo.model, "yaml", {}).get("yaml_file", "")):
        from ultralytics import RTDETR

        yolo = RTDETR(str(src))                       # what the reference itself does (extract.py:223-225)
    net = yolo.model.float().fuse().eval()
    sd = {k: v.detach().float().contiguous() for k, v in net.state_dict().items()
          if v.dtype.is_floating_point and "dfl" not in k and "num_batches_tracked" not in k}
    dec = net.model[-1]
    if type(dec).__name__ == "RTDETRDecoder":
        import torch

        layer = dec.decoder.layers[0]
        aifi = next((m for m in net.model if type(m).__name__ == "AIFI"), None)
        sd["rtdetr.meta"] = torch.tensor([float(layer.cross_attn.n_heads), float(layer.cross_attn.n_points), float(dec.num_queries),
                                          float(aifi.ma.num_heads if aifi is not None else 8)])
    save_file(sd, str(dst))
    dst.with_suffix(".names.yaml").write_text(yaml.safe_dump({int(k): str(v) for k, v in yolo.names.items()}))
    print(f"wrote {dst} ({len(sd)} tensors) and {dst.with_suffix('.names.yaml')}")


if __name__ == "__main__":
    main()
