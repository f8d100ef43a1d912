"""Which hardware queue every stream of the extract engine (+ the feeder's copy stream) lands on, for a given stream-creation
order: runs the product's track_with_model on a short .y4m under rocprofv3 and groups the kernel trace by Queue_Id.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/queue_map.py        (GTX_ENGINE_ORDER=... in the environment)
    python tools/queue_map.py --read DIR"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "geo-trax_amd"))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    f = sorted(glob.glob(sys.argv[2] + "/**/*_kernel_trace.csv", recursive=True))[-1]
    q = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        fam = ("feeder(yuv)" if "yuv420" in n else "detector" if any(k in n for k in ("conv_", "preprocess", "sppf", "head_", "nms_")) else
               "gmc" if any(k in n for k in ("lk_kernel", "response_kernel", "pyrdown")) else
               "stabilizer" if any(k in n for k in ("fast_detect", "harris", "select_kernel", "describe", "match_kernel", "ransac", "pyr_")) else "copies/other")
        q[r["Queue_Id"]][fam] += 1
    for k, v in sorted(q.items()):
        print(f"queue {k}: " + ", ".join(f"{a} {b}" for a, b in sorted(v.items())))
    sys.exit(0)

import argparse  # noqa: E402
import logging  # noqa: E402
import tempfile  # noqa: E402
from pathlib import Path  # noqa: E402

import yaml  # noqa: E402
from geotrax_amd import extract as ex  # noqa: E402
from geotrax_amd.config_utils import DEFAULT_CFG, load_config_all  # noqa: E402
from geotrax_amd.frames import bgr_to_i420, write_y4m  # noqa: E402
from geotrax_amd.synth import make_scene  # noqa: E402
from geotrax_amd.weights import save_weights, synthetic_yolov8  # noqa: E402

logger = logging.getLogger("queue_map")
logger.setLevel(logging.ERROR)
root = Path(tempfile.mkdtemp(prefix="gtx_qmap_"))
sc = make_scene(seed=0, h=2160, w=3840)
fr = [sc.render(t, 150) for t in range(3)]
pl = [bgr_to_i420(f) for f in fr]
write_y4m(root / "clip.y4m", [fr[0]] + [pl[k % 3] for k in range(1, 40)])
save_weights(synthetic_yolov8(seed=0, nc=4), root / "w.safetensors")
cfg = yaml.safe_load(DEFAULT_CFG.read_text())
cfg["ultralytics"].update(imgsz=1920, max_det=300, conf=0.25, classes=[0, 1, 2, 3], agnostic_nms=True, rect=False)
cfg["tracker"]["active"] = os.environ.get("QMAP_TRACKER", "bytetrack")
cfg["extraction"]["model"] = str(root / "w.safetensors")
(root / "cfg.yaml").write_text(yaml.safe_dump(cfg))
a = argparse.Namespace(source=str(root / "clip.y4m"), cfg=root / "cfg.yaml", output_folder=None, log_path=None, verbose=False, model=None, class_names=None,
                       conf=None, classes=None, cut_frame_left=None, cut_frame_right=None, interpolate=None)
model = ex.load_detector(a, logger)
config = load_config_all(a, logger, model_names=model.names)
ex.track_with_model(model, config, logger)
print("order", os.environ.get("GTX_ENGINE_ORDER", "(default)"), getattr(model, "last_run", None))
