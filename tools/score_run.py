#!/usr/bin/env python3
"""Scores this build's extract output against files produced by the real reference on the same video
(an operator runs `geotrax extract <video>` where ultralytics / stabilo / OpenCV are installed and hands over
results/<stem>.txt and results/<stem>_vid_transf.txt). Metrics: SURVEY.md section 8d (geotrax_amd.agreement).

    python tools/score_run.py OURS.txt REFERENCE.txt [--transforms OURS_vid_transf.txt REFERENCE_vid_transf.txt]
                              [--width 3840 --height 2160] [--stabilized]

With no arguments it scores the golden file of the reference against itself (a smoke test of the tool).
Exit code 1 when the proposed bars are missed: F1@0.5 < 0.95, median centre error > 1 px, homography grid difference > 1 px."""
import argparse
import gzip
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "geo-trax_amd"))
from geotrax_amd import agreement as A  # noqa: E402


def load(p):
    p = Path(p)
    return np.loadtxt(gzip.open(p) if p.suffix == ".gz" else p, delimiter=",", ndmin=2)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("ours", nargs="?", default=str(ROOT / "tests/golden/U_video_cut.txt.gz"))
    ap.add_argument("reference", nargs="?", default=str(ROOT / "tests/golden/U_video_cut.txt.gz"))
    ap.add_argument("--transforms", nargs=2, default=None, metavar=("OURS", "REFERENCE"))
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--stabilized", action="store_true", help="compare the stabilized boxes (columns 6-9) instead of the raw ones (2-5)")
    a = ap.parse_args()
    ours, ref = load(a.ours), load(a.reference)
    cols = (6, 10) if a.stabilized else (2, 6)
    out = {"boxes": A.box_agreement(ours, ref, cols), "tracks_ours": A.track_statistics(ours), "tracks_reference": A.track_statistics(ref)}
    ok = out["boxes"]["f1@0.5"] >= 0.95 and out["boxes"]["centre_error_px"]["median"] <= 1.0
    if a.transforms:
        to, tr = load(a.transforms[0]), load(a.transforms[1])
        out["homographies"] = A.homography_agreement(to, tr, (a.width, a.height))
        out["envelope_ours"], out["envelope_reference"] = A.homography_envelope(to), A.homography_envelope(tr)
        ok = ok and out["homographies"]["grid_diff_px"]["max"] <= 1.0
    out["within_proposed_bars"] = bool(ok)
    print(json.dumps(out, indent=1))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
