"""The reference's own `centered_pair_predictions.slp` (1100 frames, 24 nodes, 2-5 predicted instances per frame) as compact
arrays for the tracker tests: tests/golden/slp/centered_pair_predictions.arrays.npz. The reference's tracking integration tests
(tests/nn/test_tracking_integration.py:27-200) re-track exactly this file. Run once, in this container, under an interpreter
with h5py:

    /opt/conda/bin/python3.9 tools/make_golden_tracks.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import slp_io  # noqa: E402

SRC = "/root/reference/tests/data/hdf5_format_v1/centered_pair_predictions.slp"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "slp", "centered_pair_predictions.arrays.npz")
slp_io.read(SRC, "/tmp/_cpp.npz")
d = np.load("/tmp/_cpp.npz", allow_pickle=True)
fr, inst, pp = d["frames"], d["instances"], d["pred_points"]
order = np.argsort(fr["frame_idx"], kind="stable")
F, I, N = len(fr), int((fr["instance_id_end"] - fr["instance_id_start"]).max()), 24
pts = np.full((F, I, N, 2), np.nan, np.float32)
vals = np.full((F, I, N), np.nan, np.float32)
scores = np.full((F, I), np.nan, np.float32)
old_track = np.full((F, I), -1, np.int32)
n_valid = np.zeros((F,), np.int32)
frame_idx = np.zeros((F,), np.int64)
for k, f in enumerate(order):
    a, b = int(fr["instance_id_start"][f]), int(fr["instance_id_end"][f])
    frame_idx[k] = fr["frame_idx"][f]
    n_valid[k] = b - a
    for i, j in enumerate(range(a, b)):
        p = pp[int(inst["point_id_start"][j]):int(inst["point_id_end"][j])]
        vis = p["visible"].astype(bool)
        pts[k, i, :, 0] = np.where(vis, p["x"], np.nan)
        pts[k, i, :, 1] = np.where(vis, p["y"], np.nan)
        vals[k, i] = p["score"]
        scores[k, i] = inst["score"][j]
        old_track[k, i] = inst["track"][j]
np.savez_compressed(OUT, instance_peaks=pts, instance_peak_vals=vals, instance_scores=scores, n_valid=n_valid, frame_ind=frame_idx,
                    file_tracks=old_track)
print(OUT, os.path.getsize(OUT), "bytes;", F, "frames, up to", I, "instances")
