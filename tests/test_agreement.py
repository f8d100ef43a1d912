"""Agreement metrics (geotrax_amd.agreement, tools/score_run.py) on the reference's golden clip:
the file scores perfectly against itself, degrades as expected under known perturbations, and its homography
table defines the envelope the synthetic-clip tests hold this build's stabilizer to."""
import gzip
import subprocess
import sys
from pathlib import Path

import numpy as np

G = Path(__file__).parent / "golden"
ROOT = Path(__file__).resolve().parent.parent


def _gold():
    return np.loadtxt(gzip.open(G / "U_video_cut.txt.gz"), delimiter=","), np.loadtxt(G / "U_video_cut_vid_transf.txt", delimiter=",")


def test_golden_scores_perfectly_against_itself_and_degrades_under_perturbation():
    from geotrax_amd import agreement as A

    t, T = _gold()
    t = t[t[:, 0] < 40]
    same = A.box_agreement(t, t)
    assert same["f1@0.5"] == 1.0 and same["f1@[.5:.95]"] == 1.0 and same["centre_error_px"]["max"] == 0.0
    assert same["id_fragmentations"] == 0 and same["ref_tracks_matched"] == same["ref_tracks_with_one_id_here"] == len(np.unique(t[:, 1]))
    rng = np.random.default_rng(0)
    p = t.copy()
    p[:, 2:4] += rng.normal(0, 0.8, (len(p), 2))                       # sub-pixel jitter: still the same boxes
    p = p[rng.random(len(p)) > 0.05]                                    # 5 % of the boxes missing
    split = p[:, 1] == p[0, 1]
    p[split & (p[:, 0] >= 20), 1] = 9999                                # one track changes id half way
    d = A.box_agreement(p, t)
    assert 0.93 < d["recall@0.5"] < 0.97 and d["precision@0.5"] > 0.99 and d["f1@[.5:.95]"] < 1.0
    assert 0.5 < d["centre_error_px"]["median"] < 1.5 and d["id_fragmentations"] == 1
    s = A.track_statistics(t)
    assert s["tracks"] == len(np.unique(t[:, 1])) and s["rows"] == len(t) and s["length"]["max"] == 40
    h = A.homography_agreement(T, T)
    assert h["frames_common"] == 149 and h["grid_diff_px"]["max"] == 0.0
    T2 = T.copy()
    T2[:, 3] += 0.5                                                     # half a pixel of x translation
    assert abs(A.homography_agreement(T2, T)["grid_diff_px"]["max"] - 0.5) < 5e-3


def test_golden_homography_envelope():
    """149 homographies, translation grows to (2.96, 6.00) px, perspective terms ~1e-7."""
    from geotrax_amd import agreement as A

    e = A.homography_envelope(_gold()[1])
    assert e["n"] == 149 and e["det_min"] > 0.99 and e["h33_dev_max"] < 1e-12
    assert e["perspective_abs_max"] < 1e-6 and e["rotation_abs_max"] < 5e-3 and e["scale_dev_max"] < 5e-3
    assert 2.5 < e["translation_abs_max"][0] < 3.5 and 5.5 < e["translation_abs_max"][1] < 6.5 and e["translation_step_max"] < 2.5


def test_score_run_tool_smoke():
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "score_run.py"), "--transforms", str(G / "U_video_cut_vid_transf.txt"),
                        str(G / "U_video_cut_vid_transf.txt")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and '"within_proposed_bars": true' in p.stdout, p.stdout[-500:] + p.stderr[-500:]
