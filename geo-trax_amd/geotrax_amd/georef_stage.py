#!/usr/bin/env python3
"""georeference -- the stage that turns pixel tracks into georeferenced trajectories, on MI355X.

Host-side restatement of the reference stage driver geotrax/georeference.py:109-202 (same function names, argument
meaning, file conventions and log lines), chaining what this build has underneath:

    tracks txt (results/<stem>.txt, 14/15 columns)         get_tracking_data          :205-239
    flight-log timestamps (<video>.csv)                      get_timestamps             :242-268
    reference frame + fps                                    get_video_data             :271-297   (frame source, no cv2)
    orthophoto folder / parameters / lanes / master frame    get_ortho_* / get_master_* :300-516
    reference -> (master ->) orthophoto homography           estimate_homography        utils/registration.py:21-95   [HIP: RootSIFT + 2-NN + RANSAC]
    master -> orthophoto cache with an MD5 guard             get_master_to_ortho_homography :519-566
    per-row frame px -> ortho px -> lat/lon -> local metres  apply_homography/ortho2geo/geo2local :599-628         [HIP: gtx_op_georef_points]
    dimensions, visibility, kinematics, lanes, CSV           geotrax_amd.georeference   :651-876
    <stem>.csv + <stem>_geo_transf.txt                       save_georeferenced_data / save_homography :869-889

Images are read with Pillow (PNG / TIFF) or numpy (.npy); BGR like cv2.imread.

Usage:  python -m geotrax_amd.georef_stage <source> [options]      (same flags as `geotrax georeference`)
"""
from __future__ import annotations

import argparse
import hashlib
import logging
import sys
from pathlib import Path

import numpy as np

from . import georeference as G
from .config_utils import backfill_args_from_config, load_config_all
from .extract import add_common_args, get_output_dir, setup_logger
from .frames import open_source
from .registration import estimate_homography

DEFAULT_FPS = 29.97           # the reference reads CAP_PROP_FPS; the footage it is built for is 29.97 fps (README.md:382)


# --------------------------------------------------------------------------- file conventions (file_utils.py)

def build_result_path(source: Path, result_type: str, out_cfg: dict | None = None) -> Path:
    """file_utils.build_result_path (:43-71) for the result types this stage touches."""
    cfg = out_cfg or {}
    out_dir = get_output_dir(source, cfg)
    stem = Path(source).stem
    post = {"processed": cfg.get("tracks_postfix", ""), "video_transformations": cfg.get("stab_transform_postfix", "_vid_transf"),
            "geo_transformations": cfg.get("geo_transform_postfix", "_geo_transf"), "georeferenced": cfg.get("georeferenced_postfix", "")}[result_type]
    return out_dir / f"{stem}{post}.{'csv' if result_type == 'georeferenced' else 'txt'}"


def detect_delimiter(filepath: Path, lines_to_check: int = 5) -> str:
    counts = {",": 0, " ": 0, "\t": 0}
    with open(filepath, "r") as f:
        for _ in range(lines_to_check):
            line = f.readline()
            if not line:
                break
            for d in counts:
                counts[d] += line.count(d)
    return max(counts, key=lambda k: counts[k])


def determine_location_id(source: Path, logger: logging.Logger | None = None) -> str:
    """First run of letters of the file name, '_' '-' and digits end it (file_utils.py:102-130): 'A1.mp4' -> 'A'."""
    loc = []
    for ch in Path(source).stem:
        if ch.isalpha():
            loc.append(ch)
        elif loc and (ch in "_-" or ch.isdigit()):
            break
    loc = "".join(loc)
    if not loc:
        (logger.error if logger else print)(f"Error: Failed to extract location ID from the source filename {source}.")
        sys.exit(1)
    if logger:
        logger.info(f"Detected location ID: '{loc}' from the source filename {Path(source).name}.")
    return loc


def get_ortho_folder(source: Path, ortho_folder: Path | None, logger: logging.Logger) -> Path:
    """file_utils.get_ortho_folder (:133-165): given folder, else 'ORTHOPHOTOS' beside the 'PROCESSED' / 'DATASET' ancestor."""
    if ortho_folder is None:
        p = Path(source).resolve().parent
        while p != p.parent and p.name not in ("PROCESSED", "DATASET"):
            p = p.parent
        if p.name not in ("PROCESSED", "DATASET"):
            logger.critical(f"Failed to find the orthophoto folder for source '{source}'. Please either provide a custom path using the "
                            "--ortho-folder argument or ensure that the default folder structure is in place.")
            sys.exit(1)
        ortho_folder = p.parent / "ORTHOPHOTOS"
    ortho_folder = Path(ortho_folder)
    if not ortho_folder.exists():
        logger.critical(f"Orthophoto folder '{ortho_folder}' not found. Use the '--ortho-folder' argument to provide a custom path or ensure the default folder structure.")
        sys.exit(1)
    return ortho_folder


def imread_bgr(path: Path) -> np.ndarray:
    """cv2.imread(path) without cv2: .npy arrays as stored, everything else through Pillow, RGB -> BGR."""
    path = Path(path)
    if path.suffix.lower() == ".npy":
        a = np.load(path)
    else:
        from PIL import Image

        Image.MAX_IMAGE_PIXELS = None                       # orthophoto cut-outs are 15 000 px wide (default.yaml:154)
        with Image.open(path) as im:
            a = np.asarray(im.convert("RGB"))[:, :, ::-1]
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"'{path}': expected a colour image [h, w, 3], got {a.shape}")
    return a


def _find_image(folder: Path, stem: str) -> Path | None:
    for ext in (".png", ".npy", ".tif", ".jpg"):
        if (folder / f"{stem}{ext}").exists():
            return folder / f"{stem}{ext}"
    return None


# --------------------------------------------------------------------------- inputs

def get_tracking_data(source: Path, logger: logging.Logger, out_cfg: dict | None = None) -> tuple:
    """georeference.py:205-239."""
    path = build_result_path(source, "processed", out_cfg)
    if not path.exists():
        logger.critical(f"No tracking data found for: '{source}'. Run the extraction stage ('geotrax extract') first.")
        sys.exit(1)
    try:
        tracks = np.loadtxt(path, delimiter=detect_delimiter(path), dtype=np.float64)
    except Exception as e:
        logger.critical(f"Failed to load tracking data from: '{path}' due to: {e}")
        sys.exit(1)
    if tracks.size == 0 or tracks.ndim != 2:
        logger.critical(f"No valid tracking data found in: '{path}'.")
        sys.exit(1)
    if tracks.shape[1] < 14:
        logger.critical(f"Invalid tracking data format in: '{path}'. Expected at least 14 columns: [frame_id, vehicle_id, x_c_unstab, y_c_unstab, "
                        "w_unstab, h_unstab, x_c_stab, y_c_stab, w_stab, h_stab, class_id, confidence, vehicle_length, vehicle_width]. "
                        "Make sure you run geo-trax with stabilization enabled.")
        sys.exit(1)
    is_interp = tracks[:, 14].astype(int) if tracks.shape[1] >= 15 else None
    return (tracks[:, 1].astype("int"), tracks[:, 0].astype("int"), tracks[:, 2:6], tracks[:, 6], tracks[:, 7],
            tracks[:, 10].astype("int"), tracks[:, 12:14], is_interp)


def get_timestamps(source: Path, frame_num: np.ndarray, logger: logging.Logger) -> np.ndarray:
    """georeference.py:242-268: <video>.csv (or .CSV) with columns frame, timestamp."""
    import pandas as pd

    p = Path(source).with_suffix(".csv")
    if not p.exists() and Path(source).with_suffix(".CSV").exists():
        p = Path(source).with_suffix(".CSV")
    if not p.exists():
        logger.warning(f"No timestamp file found for: '{p}'. Timestamps will be replaced by frame numbers.")
        return np.array([])
    ts = pd.read_csv(p, index_col="frame")
    if ts.index[0] != 0:
        logger.warning("The first frame number in the timestamps file is not 0. Adjusting the timestamps.")
        ts.index = ts.index - ts.index[0]
    known = ts["timestamp"].to_dict()
    logger.info(f"Loaded timestamps from: '{p}'.")
    return np.array([known.get(int(f), "0000-00-00 00:00:00.000") for f in frame_num])


def get_video_data(source: Path, ref_frame_num: int, logger: logging.Logger) -> tuple:
    """(reference frame, (h, w), fps) (georeference.py:271-297) from the frame source."""
    try:
        reader = open_source(source)
    except Exception as e:
        logger.critical(f"Failed to open video file: '{source}' ({e}).")
        sys.exit(1)
    frame = None
    for _ in range(ref_frame_num + 1):
        ok, frame = reader.read()
        if not ok:
            frame = None
            break
    fps = float(getattr(reader, "fps", 0) or 0)
    reader.release()
    if frame is None:
        logger.critical(f"Failed to read frame {ref_frame_num} from video file: '{source}'.")
        sys.exit(1)
    if fps == 0:
        side = Path(source).with_suffix(".fps")             # a one-number sidecar for frame sources that carry no rate (.npy, image folders)
        fps = float(side.read_text().split()[0]) if side.exists() else DEFAULT_FPS
    frame = np.ascontiguousarray(frame.bgr() if hasattr(frame, "bgr") else frame, dtype=np.uint8)
    logger.info(f"Loaded reference frame {ref_frame_num} from: '{source}' with dimensions {frame.shape[:2]} and FPS {fps}.")
    return frame, frame.shape[:2], fps


def read_ortho_config_file(filepath: Path) -> np.ndarray:
    lines = [ln.strip() for ln in open(filepath, "r") if ln.strip() and not ln.strip().startswith("#")]
    return np.genfromtxt(lines, delimiter=" ")


def get_geo_params_source(geo_source, ortho_folder: Path, location_id: str, logger: logging.Logger) -> str:
    """georeference.py:372-418 (the .tif -> .png conversion of the reference is not done here: a .png or .npy must exist)."""
    choices = ["metadata-tif", "text-file", "center-text-file"]
    if geo_source is not None:
        if geo_source not in choices:
            logger.critical(f"Invalid --geo-source argument: '{geo_source}'. Use 'metadata-tif', 'text-file', or 'center-text-file'.")
            sys.exit(1)
        return geo_source
    base = ortho_folder / location_id
    tif, txt = base.with_suffix(".tif"), base.with_suffix(".txt")
    center, params = ortho_folder / f"{location_id}_center.txt", ortho_folder / "ortho_parameters.txt"
    if tif.exists() and (txt.exists() or (center.exists() and params.exists())):
        logger.error(f"Both .tif and .txt files are present for orthophoto '{base}'. Specify the source using the '--geo-source' argument.")
        sys.exit(1)
    if tif.exists():
        return "metadata-tif"
    if txt.exists() and center.exists() and params.exists():
        logger.error(f"Both '.txt' and '_center.txt' files are present for orthophoto: '{base}'. Specify the source using the '--geo-source' argument.")
        sys.exit(1)
    if txt.exists():
        return "text-file"
    if center.exists() and params.exists():
        return "center-text-file"
    logger.error(f"No georeferencing parameters found for orthophoto: '{base}'. Specify the source using the '--geo-source' argument.")
    sys.exit(1)


def get_ortho_parameters(ortho_folder: Path, location_id: str, geo_source: str, cutout_width_px, logger: logging.Logger) -> tuple:
    """(lng0, lat0, dlng, dlat, skew_x, skew_y) (georeference.py:318-369)."""
    base = ortho_folder / location_id
    skew_x = skew_y = 0.0
    if geo_source == "metadata-tif":
        from PIL import Image

        Image.MAX_IMAGE_PIXELS = None
        with Image.open(base.with_suffix(".tif")) as im:
            tags = im.tag_v2
            lng0, lat0 = tags[33922][3], tags[33922][4]
            dlng, dlat = tags[33550][0], -tags[33550][1]
            if 34264 in tags:
                skew_x, skew_y = tags[34264][1], tags[34264][2]
    elif geo_source == "text-file":
        p = read_ortho_config_file(base.with_suffix(".txt"))
        lng0, lat0, dlng, dlat = p[:4]
        if len(p) == 6:
            skew_x, skew_y = p[4:6]
    elif geo_source == "center-text-file":
        cx, cy = read_ortho_config_file(ortho_folder / f"{location_id}_center.txt")[:2]
        img = _find_image(ortho_folder, location_id)
        if img is None:
            logger.critical(f"Orthophoto file '{base}.png' not found.")
            sys.exit(1)
        ortho_w = imread_bgr(img).shape[1]
        half = (ortho_w if cutout_width_px is None else cutout_width_px) // 2
        p = read_ortho_config_file(ortho_folder / "ortho_parameters.txt")
        lngs, lats, dlng, dlat = p[:4]
        if len(p) == 6:
            skew_x, skew_y = p[4:6]
        lng0 = lngs + (cx - half) * dlng + (cy - half) * skew_x
        lat0 = lats + (cy - half) * dlat + (cx - half) * skew_y
        if cutout_width_px is not None and cutout_width_px != ortho_w:
            s = cutout_width_px / ortho_w
            dlng, dlat, skew_x, skew_y = dlng * s, dlat * s, skew_x * s, skew_y * s
    else:
        logger.error(f"Invalid geo_source: '{geo_source}'.")
        sys.exit(1)
    logger.info(f"Loaded orthophoto parameters from a '{geo_source}' for orthophoto: '{location_id}'.")
    return float(lng0), float(lat0), float(dlng), float(dlat), float(skew_x), float(skew_y)


def get_orthophoto(ortho_folder: Path, location_id: str, logger: logging.Logger) -> np.ndarray:
    img = _find_image(ortho_folder, location_id)
    if img is None:
        logger.critical(f"Orthophoto file '{ortho_folder / (location_id + '.png')}' not found.")
        sys.exit(1)
    a = imread_bgr(img)
    logger.info(f"Loaded orthophoto from '{img}' with dimensions: {a.shape}.")
    return a


def get_road_section_lane_geometry(ortho_folder: Path, segmentation_folder, location_id: str, logger: logging.Logger):
    import pandas as pd

    p = (Path(segmentation_folder) if segmentation_folder else ortho_folder / "segmentations") / f"{location_id}.csv"
    if p.exists():
        logger.info(f"Loaded road section and lane number geometry from: '{p}'.")
        return pd.read_csv(p).iloc[:, :10]
    logger.warning(f"No segmentation file found for: '{p}'. Road section and lane number will not be assigned.")
    return pd.DataFrame()


def get_master_frame(ortho_folder: Path, master_folder, location_id: str, logger: logging.Logger) -> np.ndarray:
    folder = Path(master_folder) if master_folder else ortho_folder / "master_frames"
    img = _find_image(folder, location_id)
    if img is None:
        logger.error(f"Master frame file '{folder / (location_id + '.png')}' not found. If you do not want to use a master frame, use the '--no-master' option.")
        sys.exit(1)
    logger.info(f"Loaded master frame from: '{img}' to act as an intermediate frame.")
    return imread_bgr(img)


# --------------------------------------------------------------------------- homographies

def compute_hash(image: np.ndarray) -> str:
    return hashlib.md5(np.ascontiguousarray(image).tobytes()).hexdigest()


def compute_homography(img_src, img_dst, src_dst: tuple, logger: logging.Logger, ctx=None, **matching) -> tuple:
    """georeference.py:569-596: estimate_homography + the stats line; no model -> exit."""
    H, inliers, n_matches, n_kp = estimate_homography(img_src, img_dst, logger, ctx=ctx, **matching)
    if H is None:
        sys.exit(1)
    stats = f"Keypoints in {src_dst[0]} frame: {n_kp[0]}, in {src_dst[1]}: {n_kp[1]}. Inliers: {inliers} out of {n_matches} matches"
    (logger.warning if inliers < 50 else logger.info)(stats)
    return H, stats


def get_master_to_ortho_homography(master_frame, ortho_folder: Path, master_folder, location_id: str, recompute: bool, matching: dict,
                                   logger: logging.Logger, ctx=None) -> np.ndarray:
    """Cached next to the master frame with the MD5 of the master image; recomputed when it changed (georeference.py:519-566)."""
    path = (Path(master_folder) if master_folder else ortho_folder / "master_frames") / f"{location_id}.txt"
    current = compute_hash(master_frame)
    if path.exists() and not recompute:
        try:
            lines = path.read_text().splitlines()
            H = np.array([float(v) for v in lines[0].split(",")]).reshape(3, 3)
            saved = lines[3].strip().split(": ")[1]
        except Exception as e:
            logger.error(f"Failed to load 'master -> orthophoto' homography from '{path}' due to: {e}")
            sys.exit(1)
        if saved == current:
            logger.info(f"Loaded 'master -> orthophoto' homography from: '{path}'.")
            return H
        logger.warning("Master frame has changed. Recomputing 'master -> orthophoto' homography.")
    H, stats = compute_homography(master_frame, get_orthophoto(ortho_folder, location_id, logger), ("master", "ortho"), logger, ctx=ctx, **matching)
    try:
        with open(path, "w") as f:
            np.savetxt(f, H.reshape(1, -1), fmt="%.20g", delimiter=",")
            f.write("\n# Hash of the master frame\n")
            f.write(f"Hash: {current}\n")
            f.write("\n# Image matching stats\n")
            f.write(f"Stats: {stats}\n")
    except Exception as e:
        logger.error(f"Failed to save 'master -> orthophoto' homography to '{path}' due to: {e}")
        sys.exit(1)
    logger.info(f"Computed and saved 'master -> orthophoto' homography to: '{path}'.")
    return H


# --------------------------------------------------------------------------- the stage

def georeference_tracks(track_id, frame_num, bbox_unstab, x_stab, y_stab, class_id, veh_dim_px, is_interpolated, timestamps, frame_size, fps,
                        H_ref2ortho, ortho_params, ortho_segmentation, config: dict, logger: logging.Logger, ctx=None):
    """Everything after the homography is known (georeference.py:172-194): the per-row chain on the GPU, then the
    per-track host arithmetic; -> the formatted DataFrame."""
    tr = config["transformation"]
    src_crs, dst_crs = tr["source_crs"], tr["target_crs"]
    out = G.transform_points(x_stab, y_stab, H_ref2ortho, ortho_params, src_crs, dst_crs, ctx=ctx)        # gtx_op_georef_points
    x_o, y_o, lat, lon, x_l, y_l = (out[k] for k in ("ortho_x", "ortho_y", "latitude", "longitude", "x_local", "y_local"))
    dim_real = G.convert_dimensions(track_id, veh_dim_px, frame_size, H_ref2ortho, ortho_params, src_crs, dst_crs)
    flt = config["filtering"]
    vis = G.calculate_visibility(track_id, bbox_unstab, frame_size, flt["visibility_margin"])
    speed, accel = G.compute_kinematics(track_id, frame_num, x_l, y_l, vis, fps, flt["filter_type"], flt["kernel_size"], is_interpolated=is_interpolated)
    section, lane = G.assign_road_section_lane(x_o, y_o, ortho_segmentation)
    return G.create_and_format_georeferenced_df(track_id, timestamps, frame_num, x_o, y_o, x_l, y_l, lat, lon, dim_real, class_id, speed, accel,
                                                section, lane, vis, flt["min_traj_length"], is_interpolated, logger=logger)


def georeference(args: argparse.Namespace, logger: logging.Logger, ctx=None) -> None:
    """Georeference the tracking data of one video using orthophotos (georeference.py:109-202)."""
    full = load_config_all(args, logger, needs_model=False)
    config = full["georef"]
    gproc = config["processing"]
    folders = full["main"].get("input", {}) or {}
    out_raw = full["main"].get("output", {}) or {}
    backfill_args_from_config(args, {
        "ref_frame": gproc["ref_frame"], "recompute": gproc["recompute"], "geo_source": gproc["geo_source"], "no_master": not gproc["use_master"],
        "ortho_folder": Path(folders["ortho_folder"]) if folders.get("ortho_folder") else None,
        "master_folder": Path(folders["master_folder"]) if folders.get("master_folder") else None,
        "segmentation_folder": Path(folders["segmentation_folder"]) if folders.get("segmentation_folder") else None,
        "output_folder": out_raw.get("folder", "results"),
    })
    out_cfg = {**out_raw, "folder": args.output_folder}
    source = Path(args.source)

    location_id = determine_location_id(source, logger)
    track_id, frame_num, bbox_unstab, x_stab, y_stab, class_id, veh_dim_px, is_interp = get_tracking_data(source, logger, out_cfg)
    timestamps = get_timestamps(source, frame_num, logger)
    reference_frame, frame_size, fps = get_video_data(source, args.ref_frame, logger)
    ortho_folder = get_ortho_folder(source, args.ortho_folder, logger)
    geo_source = get_geo_params_source(args.geo_source, ortho_folder, location_id, logger)
    ortho_params = get_ortho_parameters(ortho_folder, location_id, geo_source, config["transformation"]["cutout_width_px"], logger)
    segmentation = get_road_section_lane_geometry(ortho_folder, args.segmentation_folder, location_id, logger)

    matching = {k: v for k, v in config["matching"].items()}
    if args.no_master:
        ortho = get_orthophoto(ortho_folder, location_id, logger)
        H_ref2ortho = compute_homography(reference_frame, ortho, ("reference", "ortho"), logger, ctx=ctx, **matching)[0]
    else:
        master = get_master_frame(ortho_folder, args.master_folder, location_id, logger)
        H_ref2master = compute_homography(reference_frame, master, ("reference", "master"), logger, ctx=ctx, **matching)[0]
        H_master2ortho = get_master_to_ortho_homography(master, ortho_folder, args.master_folder, location_id, args.recompute, matching, logger, ctx=ctx)
        H_ref2ortho = np.dot(H_master2ortho, H_ref2master)

    df = georeference_tracks(track_id, frame_num, bbox_unstab, x_stab, y_stab, class_id, veh_dim_px, is_interp, timestamps, frame_size, fps,
                             H_ref2ortho, ortho_params, segmentation, config, logger, ctx=ctx)
    G.save_georeferenced_data(build_result_path(source, "georeferenced", out_cfg), df, logger)
    G.save_homography(build_result_path(source, "geo_transformations", out_cfg), H_ref2ortho, logger)


def add_georeferencing_args(group) -> None:
    """Same flags, spelling and defaults as the reference (georeference.py:892-906)."""
    group.add_argument("--ortho-folder", "-orf", type=Path, default=None)
    group.add_argument("--geo-source", "-gs", choices=["metadata-tif", "text-file", "center-text-file"], default=None)
    group.add_argument("--ref-frame", "-rf", type=int, default=None)
    group.add_argument("--no-master", "-nm", action="store_const", const=True, default=None)
    group.add_argument("--master-folder", "-mf", type=Path, default=None)
    group.add_argument("--recompute", "-r", action="store_const", const=True, default=None)
    group.add_argument("--segmentation-folder", "-osf", type=Path, default=None)


def parse_cli_args(argv=None) -> argparse.Namespace:
    parser = argparse.ArgumentParser(prog="geotrax georeference", description="Georeferencing the tracking data using orthophotos.")
    parser.add_argument("source", type=Path, help="Path to the input video / frame source.")
    add_common_args(parser.add_argument_group("Optional arguments"))
    add_georeferencing_args(parser.add_argument_group("Georeferencing arguments"))
    return parser.parse_args(argv)


def main(argv=None) -> None:
    args = parse_cli_args(argv)
    georeference(args, setup_logger("geotrax_amd.georeference", args.verbose, args.log_path))


if __name__ == "__main__":
    main()
