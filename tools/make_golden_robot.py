"""Extracts the reference-held END-TO-END pin of the network path into tests/golden/robot.npz (run once, in this container,
under the interpreter that has h5py + Pillow):

    /opt/conda/bin/python3.9 tools/make_golden_robot.py

What the reference holds (tests/nn/test_inference.py:592-610, `test_single_instance_predictor`): the trained model
`tests/data/models/minimal_robot.UNet.single_instance` run on `tests/data/slp_hdf5/small_robot_minimal.slp` must land within
`atol=10` px of that file's user-labelled points. The labelled frames are frames 0 and 79 of `small_robot.mp4` (H.264, not
decodable offline), but frame 0 .. 2 of the same video are also held as `tests/data/videos/robot{0,1,2}.jpg` (the fixtures of
the reference's `test_images_video`), which Pillow decodes. Stored here:

    frames   (3, 320, 560, 3) uint8, RGB order -- the decoded JPEGs (data, not code; decoding differences between JPEG
             libraries stay out of the tests because the decoded arrays are what is committed)
    gt_frame_idx, gt_points   the labelled frames' indices and (n, 2 nodes, 2) float64 points, skeleton node order
    node_names
"""
import json
import os

import h5py
import numpy as np
from PIL import Image

REF = "/root/reference/tests/data"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "robot.npz")

frames = np.stack([np.asarray(Image.open(os.path.join(REF, "videos", f"robot{i}.jpg")).convert("RGB")) for i in range(3)])
assert frames.shape == (3, 320, 560, 3) and frames.dtype == np.uint8
with h5py.File(os.path.join(REF, "slp_hdf5", "small_robot_minimal.slp"), "r") as f:
    fr, inst, pts = f["frames"][:], f["instances"][:], f["points"][:]
    md = json.loads(f["metadata"].attrs["json"])
# node order of the skeleton = order of the ids in skeletons[0]["nodes"]; names from the global node list
sk_ids = [n["id"] for n in md["skeletons"][0]["nodes"]]
names = [md["nodes"][i]["name"] for i in sk_ids]
gt_idx, gt_pts = [], []
for row in fr:
    ii = inst[row["instance_id_start"]: row["instance_id_end"]]
    assert len(ii) == 1
    p = pts[ii[0]["point_id_start"]: ii[0]["point_id_end"]]
    gt_idx.append(int(row["frame_idx"]))
    gt_pts.append(np.stack([p["x"], p["y"]], axis=-1))
np.savez_compressed(OUT, frames=frames, gt_frame_idx=np.array(gt_idx, np.int64), gt_points=np.stack(gt_pts),
                    node_names=np.array(names))
print("wrote", OUT, os.path.getsize(OUT), "bytes; gt frames", gt_idx, "nodes", names)
print(np.stack(gt_pts))
