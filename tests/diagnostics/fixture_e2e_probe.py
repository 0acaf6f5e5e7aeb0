"""End-to-end agreement on the TRAINED fixture model (minimal_instance.UNet.bottomup): fp32 CPU oracle (network + post-processing)
vs the device path, same synthetic fly-like frames, at several peak thresholds.

    python tests/diagnostics/fixture_e2e_probe.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import paf_grouping as opg
from oracle import peak_finding as opf
from oracle.keras_graph import KerasGraph, preprocess
from sleap_amd.nn.engine import load_keras_npz
from sleap_amd.nn.inference import load_model
from sleap_amd.synth import render_frames

MODEL = os.path.join("tests", "golden", "models", "minimal_instance.UNet.bottomup")
frames = render_frames(6, 384, 384, n_animals=2, seed=11)[0]
cfg, w = load_keras_npz(os.path.join(MODEL, "best_model.npz"))
torch.set_num_threads(16)
cms_o, pafs_o, offs_o = KerasGraph(cfg, w)(preprocess(frames))
p = load_model(MODEL, batch_size=6, progress_reporting="none")
layer = p.inference_model.bottomup_layer
cms_d, pafs_d, offs_d = [t.cpu().numpy() for t in layer.forward_pass(frames)]
for name, a, b in (("cms", cms_d, cms_o), ("pafs", pafs_d, pafs_o), ("offsets", offs_d, offs_o)):
    print(f"{name}: max|d|/max|ref| = {np.abs(a - b).max() / np.abs(b).max():.4f}   (range of ref {b.min():.3f} .. {b.max():.3f})")
sc = opg.PAFScorer(["A", "B"], [("A", "B")], 4, oob="zero")
for thr in (0.2, 0.5, 0.8, float(np.sort(cms_o[opf.nms_mask(cms_o, -np.inf)])[::-1][12 * 6 * 2])):
    layer.peak_threshold = thr
    outs = p.predict(frames, make_labels=False)[0]
    pts, vals, si, ci = opf.find_local_peaks_with_offsets(cms_o, offs_o, thr)
    pts = pts * np.float32(2)
    o = sc.predict(pafs_o, [pts[si == b] for b in range(6)], [vals[si == b] for b in range(6)], [ci[si == b] for b in range(6)])
    same_n = same_mask = 0
    dmax, npk, nclose = 0.0, 0, 0
    for b in range(6):
        n = int(outs["n_valid"][b])
        if n != len(o[0][b]):
            continue
        same_n += 1
        got, want = outs["instance_peaks"][b, :n], np.asarray(o[0][b]).reshape(n, 2, 2)
        if not np.array_equal(np.isnan(got), np.isnan(want)):
            continue
        same_mask += 1
        d = np.linalg.norm(got - want, axis=-1)
        d = d[np.isfinite(d)]
        if d.size:
            dmax, npk, nclose = max(dmax, float(d.max())), npk + d.size, nclose + int((d <= 0.5).sum())
    print(f"threshold {thr:.3f}: oracle peaks/frame {len(pts) / 6:.1f}; frames with equal instance count {same_n}/6, equal node "
          f"assignment {same_mask}/6; matched peaks within 0.5 px {nclose}/{npk}, max {dmax:.3f} px")
