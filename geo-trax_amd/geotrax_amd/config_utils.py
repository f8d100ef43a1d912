"""Pipeline configuration for the extract path: one YAML -> the four dicts the stage consumes.

Restates what the extract stage needs from geotrax/utils/config_utils.py of the reference
(load_config_all :127-194, backfill_args_from_config :241-246, resolve_class_names :307-343):
same section names, same CLI-over-config precedence, the active tracker block selected by
``tracker.active``. The reference hands ultralytics a temporary tracker YAML *file*; this build's
model object takes the block itself.
"""
from __future__ import annotations

import argparse
import json
import logging
import sys
from pathlib import Path

import yaml

CFG_DIR = Path(__file__).resolve().parent / "cfg"
DEFAULT_CFG = CFG_DIR / "default.yaml"
PRESETS = ("default", "confident", "lenient", "stable")


def resolve_config_path(cfg_filepath) -> Path:
    """config_utils.resolve_config_path (:38-63): the path as given, else inside the bundled cfg directory; a
    missing '.yaml' suffix and a leading 'cfg/' (or the reference's 'geotrax/cfg/') are tolerated, so 'confident',
    'cfg/default.yaml' and 'geotrax/cfg/lenient.yaml' all resolve to the bundled presets."""
    path = Path(cfg_filepath)
    if not path.suffix:
        path = path.with_suffix(".yaml")
    candidates = [path]
    if not path.is_absolute():
        candidates.append(CFG_DIR / path.name if path.parent.name in ("", "cfg") or path.parts[-2:-1] == ("cfg",) else CFG_DIR / path)
    for c in candidates:
        if c.is_file():
            return c
    return Path(cfg_filepath)


def _merge(base: dict, over: dict) -> dict:
    out = dict(base)
    for k, v in over.items():
        out[k] = _merge(base[k], v) if isinstance(v, dict) and isinstance(base.get(k), dict) else v
    return out


def load_config(cfg_filepath, logger: logging.Logger) -> dict:
    """One YAML -> dict. A file may carry `_base: <preset or path>`: it then only lists the keys that differ from that
    file (how the bundled confident / lenient / stable presets are written here; a full file such as the reference's
    own geotrax/cfg/*.yaml has no `_base` and loads unchanged)."""
    path = resolve_config_path(cfg_filepath) if cfg_filepath else DEFAULT_CFG
    try:
        with open(path, "r") as f:
            cfg = yaml.safe_load(f) or {}
    except FileNotFoundError:
        logger.critical(f"Configuration file '{cfg_filepath}' not found.")
        sys.exit(1)
    base = cfg.pop("_base", None)
    if base is not None:
        cfg = _merge(load_config(base, logger), cfg)
    return cfg


def backfill_args_from_config(args: argparse.Namespace, mapping: dict) -> None:
    for name, value in mapping.items():
        if getattr(args, name, None) is None:
            setattr(args, name, value)


def _class_mapping(value, logger):
    mapping = None
    if isinstance(value, dict):
        mapping = value
    elif isinstance(value, list):
        if len(value) == 1 and Path(value[0]).is_file():
            return _class_mapping(value[0], logger)
        mapping = {}
        for tok in value:
            if "=" not in tok:
                logger.error(f"Invalid --class-names entry '{tok}'. Expected ID=NAME (e.g. 0=car) or a file path.")
                return None
            k, v = tok.split("=", 1)
            mapping[k] = v
    else:
        p = Path(value)
        if not p.is_file():
            logger.error(f"Class names file '{p}' not found.")
            return None
        mapping = json.loads(p.read_text()) if p.suffix.lower() == ".json" else yaml.safe_load(p.read_text())
    if not isinstance(mapping, dict) or not mapping:
        return None
    try:
        return {int(k): str(v) for k, v in mapping.items()}
    except (ValueError, TypeError):
        logger.error(f"Class names override '{value}' has non-integer keys.")
        return None


def resolve_class_names(model_names, cli_value, cfg_value, classes, logger):
    """Precedence CLI > config > model > integer fallback; returns (mapping, source_label)."""
    for label, value in (("cli", cli_value), ("config", cfg_value)):
        if value is not None:
            m = _class_mapping(value, logger)
            if m is not None:
                return m, label
    if model_names:
        return dict(model_names), "model"
    ids = classes if classes else range(100)
    logger.warning("No class-name mapping found (CLI, config, or model); falling back to integer class IDs.")
    return {int(i): str(int(i)) for i in ids}, "fallback"


def load_config_all(args: argparse.Namespace, logger: logging.Logger, model_names: dict | None = None, needs_model: bool = True) -> dict:
    """config_utils.load_config_all (:127-194). needs_model=False (the georeference stage): no tracker block, model or
    class names are resolved, so a missing model does not stop a stage that never uses it."""
    full = load_config(getattr(args, "cfg", None), logger)
    if not needs_model:
        main = {k: v for k, v in full.items() if k not in ("tracker", "stabilo", "ultralytics", "georef")}
        main.update(class_names={}, class_names_source=None, model_configured=None, tracker_active=None, tracker_params={}, args=args)
        return {"main": main, "stabilo": dict(full.get("stabilo", {})), "ultralytics": dict(full.get("ultralytics", {})), "georef": full.get("georef", {})}
    tracker = full.get("tracker", {})
    stabilo = dict(full.get("stabilo", {}))
    ultra = dict(full.get("ultralytics", {}))
    georef = full.get("georef", {})
    main = {k: v for k, v in full.items() if k not in ("tracker", "stabilo", "ultralytics", "georef")}
    main["tracker"] = tracker

    active = tracker.get("active")
    if active is None or active not in tracker:
        logger.critical(f"Active tracker '{active}' has no parameter block in the 'tracker' section.")
        sys.exit(1)
    ultra["tracker"] = dict(tracker[active])
    extraction = full.get("extraction", {})
    raw_model = getattr(args, "model", None)
    if isinstance(raw_model, list):
        raw_model = " ".join(raw_model)
    model_ref = raw_model or extraction.get("model") or ultra.get("model")
    main["model_configured"] = str(model_ref)
    ultra["model"] = str(model_ref)
    main["class_names"], main["class_names_source"] = resolve_class_names(
        model_names, getattr(args, "class_names", None), extraction.get("class_rename"), ultra.get("classes"), logger)
    main["tracker_active"] = active
    main["tracker_params"] = tracker.get(active, {})
    main["args"] = args
    for arg in ("classes", "conf", "show"):
        value = getattr(args, arg, None)
        if value is not None:
            ultra[arg] = value
            logger.info(f"The default ultralytics value for {arg} has been updated to the provided CLI argument: {value}.")
    return {"main": main, "stabilo": stabilo, "ultralytics": ultra, "georef": georef}
