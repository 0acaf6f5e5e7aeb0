"""Regenerate tests/golden/models/*.npz (+ training_config.json) from the reference's test
fixture models. Run HERE (needs /root/reference and an h5py interpreter):

    /opt/conda/bin/python3.9 tools/make_golden_models.py

The .npz files hold the Keras graph JSON and float32 weights exactly as stored in
`tests/data/models/<name>/best_model.h5` of the reference checkout (data, not code).
"""
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(__file__))
from h5_extract import extract

REF = "/root/reference/tests/data/models"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "models")
NAMES = [
    "minimal_instance.UNet.bottomup",
    "minimal_instance.UNet.centroid",
    "minimal_instance.UNet.centered_instance",
    "minimal_robot.UNet.single_instance",
    "min_tracks_2node.UNet.bottomup_multiclass",
]
for n in NAMES:
    d = os.path.join(OUT, n)
    os.makedirs(d, exist_ok=True)
    extract(os.path.join(REF, n, "best_model.h5"), os.path.join(d, "best_model.npz"))
    shutil.copy(os.path.join(REF, n, "training_config.json"), os.path.join(d, "training_config.json"))
    print("wrote", d)
