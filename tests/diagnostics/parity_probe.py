"""Why do the fp32 CPU oracle and the bf16 device path disagree on some frames of the random-init benchmark model?
Runs the ORACLE post-processing on (a) the oracle's fp32 maps and (b) the device's maps for bench.py's 8 sample frames and
reports, per frame, instance counts and how close the decisive PAF line scores sit to the min_line_scores cut (0.25).

    python tests/diagnostics/parity_probe.py
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import paf_grouping as opg
from oracle import peak_finding as opf
from oracle.keras_graph import KerasGraph, preprocess
from sleap_amd.benchmark_model import build_benchmark_predictor
from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES, render_frames

H = W = 1024
pred, mc, weights = build_benchmark_predictor(H, W, batch_size=8, seed=0)
frames, _ = render_frames(8, H, W, n_animals=4, seed=100)
layer = pred.inference_model.bottomup_layer
cms_d, pafs_d, _ = layer.forward_pass(frames)
cms_d, pafs_d = cms_d.cpu().numpy(), pafs_d.cpu().numpy()
torch.set_num_threads(32)
cms_o, pafs_o = KerasGraph(mc, weights)(preprocess(frames))[:2]
print("network: max|d|/max|ref| cms %.4f pafs %.4f" % (np.abs(cms_d - cms_o).max() / np.abs(cms_o).max(),
                                                         np.abs(pafs_d - pafs_o).max() / np.abs(pafs_o).max()))
sc = opg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8, oob="zero")


def post(cms, pafs):
    pts, vals, si, ci = opf.find_local_peaks(cms, 0.2, "integral", 5)
    pts = pts * np.float32(4)
    n = cms.shape[0]
    return sc.predict(pafs, [pts[si == b] for b in range(n)], [vals[si == b] for b in range(n)], [ci[si == b] for b in range(n)]), \
        [int((si == b).sum()) for b in range(n)]


(ro, npk_o), (rd, npk_d) = post(cms_o, pafs_o), post(cms_d, pafs_d)
for b in range(8):
    lo, ld = np.asarray(ro[5][b]), np.asarray(rd[5][b])
    near_o = int((np.abs(lo - 0.25) < 0.03).sum())
    print(f"frame {b}: peaks oracle/device {npk_o[b]}/{npk_d[b]}  instances {len(ro[0][b])}/{len(rd[0][b])}  "
          f"candidates {lo.size}/{ld.size}  line scores within 0.03 of the 0.25 cut: {near_o}  "
          f"above cut {int((lo > 0.25).sum())}/{int((ld > 0.25).sum())}")
    if lo.size == ld.size and lo.size:
        d = np.abs(lo - ld)
        print(f"         |line score delta| max {d.max():.4f} mean {d.mean():.4f}; flips across the cut: "
              f"{int(((lo > 0.25) != (ld > 0.25)).sum())}")
