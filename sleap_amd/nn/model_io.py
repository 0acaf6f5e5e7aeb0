"""Reading SLEAP model folders: `training_config.json` + `best_model.h5` (inference.py:132-144, 3204-3209).

The config is consumed as plain JSON (fields listed in SURVEY.md §8b). The Keras HDF5 is read with
h5py when the running interpreter has it; otherwise `sleap_amd/nn/_h5_extract.py` (part of the package: no
repository checkout needed) is run once under an interpreter that does (env `SLEAP_AMD_H5_PYTHON`, default /opt/conda/bin/python3.9) and the result is
cached as `best_model.npz` next to the HDF5 (or in `$SLEAP_AMD_CACHE` if the folder is read-only).
A pre-extracted `best_model.npz` in the folder is used directly.
"""
import hashlib
import json
import math
import os
import re
import subprocess
from typing import Dict, Tuple

import numpy as np

HEAD_TYPES = ("single_instance", "centroid", "centered_instance", "multi_instance", "multi_class_bottomup",
              "multi_class_topdown")


def model_dir(path: str) -> str:
    if os.path.isfile(path) and path.endswith(".json"):
        return os.path.dirname(path)
    return path


def _strip_json_comments(text: str) -> str:
    # training_job.py:93-124 runs jsmin before json.loads
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"(^|\s)//[^\n]*", r"\1", text)


def load_training_config(path: str) -> dict:
    p = path if (os.path.isfile(path) and path.endswith(".json")) else os.path.join(path, "training_config.json")
    with open(p, "r") as f:
        text = f.read()
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        return json.loads(_strip_json_comments(text))


def head_type(cfg: dict) -> str:
    """`cfg.model.heads.which_oneof_attrib_name()` (inference.py:230-232)."""
    heads = cfg["model"]["heads"]
    set_ = [k for k in HEAD_TYPES if heads.get(k) is not None]
    if len(set_) != 1:
        raise ValueError(f"exactly one head type must be set, found {set_}")
    return set_[0]


def backbone_type(cfg: dict) -> str:
    bb = cfg["model"]["backbone"]
    set_ = [k for k, v in bb.items() if v is not None]
    if len(set_) != 1:
        raise ValueError(f"exactly one backbone must be set, found {set_}")
    return set_[0]


def maximum_stride(cfg: dict) -> int:
    """`Model.from_config(cfg.model).maximum_stride` (inference.py:3119-3140 uses it as pad_to_stride)."""
    bt = backbone_type(cfg)
    bb = cfg["model"]["backbone"][bt]
    if bt in ("unet", "leap", "hourglass", "resnet"):
        return int(bb["max_stride"])
    if bt == "pretrained_encoder":
        return int(bb.get("encoder_features_stride", bb.get("max_stride", 32)))
    raise ValueError(bt)


def _h5_python():
    return os.environ.get("SLEAP_AMD_H5_PYTHON", "/opt/conda/bin/python3.9")


def load_keras_model(folder: str) -> Tuple[dict, Dict[str, np.ndarray]]:
    """-> (model_config dict as stored in the HDF5 attr `model_config`, {"<layer>/<weight>": float32 array})."""
    npz = os.path.join(folder, "best_model.npz")
    h5 = os.path.join(folder, "best_model.h5")
    # an extracted copy is only trusted while it is at least as new as the .h5 beside it (a model retrained or overwritten
    # in the same folder must not be served from stale weights)
    stale = os.path.exists(npz) and os.path.exists(h5) and os.path.getmtime(h5) > os.path.getmtime(npz)
    if stale or not os.path.exists(npz):
        if not os.path.exists(h5):
            raise FileNotFoundError(f"neither best_model.h5 nor best_model.npz in {folder}")
        target = _cache_path(folder, h5)
        try:
            import h5py  # noqa: F401
        except ImportError:
            script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_h5_extract.py")
            py = _h5_python()
            if not os.path.exists(py):
                raise ImportError(f"reading {h5} needs h5py: not importable here, and SLEAP_AMD_H5_PYTHON ({py}) does not exist -- "
                                  "install h5py, point SLEAP_AMD_H5_PYTHON at an interpreter that has it, or put a pre-extracted "
                                  "best_model.npz (python sleap_amd/nn/_h5_extract.py best_model.h5 best_model.npz) beside it")
            subprocess.check_call([py, script, h5, target])
        else:
            from ._h5_extract import extract

            extract(h5, target)
        npz = target
    z = np.load(npz)
    cfg = json.loads(bytes(z["__model_config__"]).decode("utf-8"))
    return cfg, {k: z[k] for k in z.files if k != "__model_config__"}


def _cache_path(folder, h5):
    if os.access(folder, os.W_OK):
        return os.path.join(folder, "best_model.npz")
    cache = os.environ.get("SLEAP_AMD_CACHE", os.path.join(os.path.expanduser("~"), ".cache", "sleap_amd"))
    os.makedirs(cache, exist_ok=True)
    key = hashlib.sha1((os.path.abspath(h5) + str(os.path.getmtime(h5))).encode()).hexdigest()[:16]
    return os.path.join(cache, f"{key}.npz")
