"""Extracts the tables of the reference's own prediction files into tests/golden/slp/*.npz (run once, in this container,
under an interpreter with h5py):

    /opt/conda/bin/python3.9 tools/make_golden_slp.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import slp_io  # noqa: E402

REF = "/root/reference/tests/data"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "slp")
FILES = {"bottomup.labels_pr.val": "models/minimal_instance.UNet.bottomup/labels_pr.val.slp",
         "bottomup.labels_pr.train": "models/minimal_instance.UNet.bottomup/labels_pr.train.slp",
         "dance.labels": "slp_hdf5/dance.mp4.labels.slp",
         # round 6: the top-down models' own prediction files (frame 0 of centered_pair_low_quality.mp4: centroid confidences /
         # centered-instance peaks on ground-truth centroid crops) -- tests/test_frame0_golden.py
         "centroid.labels_pr.val": "models/minimal_instance.UNet.centroid/labels_pr.val.slp",
         "centered_instance.labels_pr.val": "models/minimal_instance.UNet.centered_instance/labels_pr.val.slp"}
os.makedirs(OUT, exist_ok=True)
for name, rel in FILES.items():
    slp_io.read(os.path.join(REF, rel), os.path.join(OUT, name + ".npz"))
    print(name)
