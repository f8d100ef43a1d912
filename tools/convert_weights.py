#!/usr/bin/env python3
"""Converts an ultralytics YOLOv8 detect checkpoint (.pt) into the flat .safetensors file +
`<name>.names.yaml` side-car that geotrax_amd.model.YOLO loads.

Run this once on a machine where `ultralytics` is installed (it is not part of this build):

    python tools/convert_weights.py geotrax_hbb_yolov8s_1920_v1.pt [out.safetensors]

Conv+BN pairs are fused with ultralytics' own `model.fuse()`; tensor names are the fused model's
state_dict keys (`model.0.conv.weight`, `model.0.conv.bias`, ... `model.22.cv3.2.2.bias`).
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
    net = yolo.model.float().fuse().eval()
    sd = {k: v.detach().float().contiguous() for k, v in net.state_dict().items()
          if v.dtype.is_floating_point and "dfl" not in k and "num_batches_tracked" not in k}
    save_file(sd, str(dst))
    dst.with_suffix(".names.yaml").write_text(yaml.safe_dump({int(k): str(v) for k, v in yolo.names.items()}))
    print(f"wrote {dst} ({len(sd)} tensors) and {dst.with_suffix('.names.yaml')}")


if __name__ == "__main__":
    main()
