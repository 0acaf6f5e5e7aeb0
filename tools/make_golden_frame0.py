"""Freeze frame 0 of the reference's centered_pair_low_quality.mp4 as decoded by the package's own intra-picture H.264 decoder,
with the user labels of that frame (the reference's `min_labels` fixture) -- run once, in this container:

    python tools/make_golden_frame0.py

Writes tests/golden/centered_pair_frame0.npz: `luma` (the decoded Y plane, what ANY conforming H.264 decoder yields), `gray`
(the frame as sleap.io.video.MediaVideo hands it to the model: libswscale's limited -> full range conversion, channel 0),
`gt_points` (tests/data/json_format_v2/minimal_instance.json: two instances x two nodes). The MP4 itself is committed next to it
(tests/golden/video/, a data file of the reference's test suite); the expected predictions are
tests/golden/slp/bottomup.labels_pr.val.npz (tools/make_golden_slp.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sleap_amd.io import _h264_intra as H  # noqa: E402

REF = "/root/reference/tests/data"
y, cb, cr, st = H.decode_intra(os.path.join(REF, "json_format_v1", "centered_pair_low_quality.mp4"), 0)
assert int(cb.min()) == int(cb.max()) == int(cr.min()) == int(cr.max()) == 128  # a grey stream
gray = H.swscale_bgr(y, cb, cr)[..., 0]
lab = json.load(open(os.path.join(REF, "json_format_v2", "minimal_instance.json")))["labels"][0]
assert lab["frame_idx"] == 0
gt = np.array([[[inst["_points"][str(n)]["x"], inst["_points"][str(n)]["y"]] for n in range(2)] for inst in lab["_instances"]], np.float64)
out = os.path.join(ROOT, "tests", "golden", "centered_pair_frame0.npz")
np.savez_compressed(out, luma=y, gray=gray, gt_points=gt)
print(out, y.shape, st, "gt", gt.tolist())
