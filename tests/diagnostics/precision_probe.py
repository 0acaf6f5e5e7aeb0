"""CPU-only: how much of the end-to-end disagreement with the fp32 network is due to 16-bit STORAGE precision, and what would
fp16 (11-bit mantissa) instead of bf16 (8-bit) buy? The oracle network is run three times on the same frames -- fp32, with
bf16 rounding at the engine's storage points (activations after every conv, weights), with fp16 rounding at the same points --
and the restated post-processing is applied to each.

    python tests/diagnostics/precision_probe.py

It then measures how well-conditioned the comparison is at all: the fp32 maps are perturbed with white noise of amplitude
eps x range and the oracle's result is compared with its own unperturbed result. (Measured: profiles/r01_precision_probe.md --
the decisions on this fixture survive eps = 1e-4, start to flip at 3e-4; fp16 storage is ~1e-3, bf16 ~1e-2 .. 4e-2.)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import paf_grouping as opg
from oracle import peak_finding as opf
from oracle.keras_graph import KerasGraph, load_npz_model, preprocess
from sleap_amd.synth import render_frames

torch.set_num_threads(int(__import__("os").environ.get("PROBE_THREADS", "8")))


def post(cms, pafs, offs, thr, scorer, stride):
    if offs is not None:
        pts, vals, si, ci = opf.find_local_peaks_with_offsets(cms, offs, thr)
    else:
        pts, vals, si, ci = opf.find_local_peaks(cms, thr, "integral", 5)
    pts = pts * np.float32(stride)
    n = cms.shape[0]
    return scorer.predict(pafs, [pts[si == b] for b in range(n)], [vals[si == b] for b in range(n)], [ci[si == b] for b in range(n)]), \
        [int((si == b).sum()) for b in range(n)]


def compare(ref, got, n_nodes):
    (ro, npo), (rg, npg) = ref, got
    same_n = same_mask = npk = nclose = 0
    for b in range(len(npo)):
        a, c = np.asarray(ro[0][b]).reshape(-1, n_nodes, 2), np.asarray(rg[0][b]).reshape(-1, n_nodes, 2)
        if len(a) != len(c):
            continue
        same_n += 1
        if not np.array_equal(np.isnan(a), np.isnan(c)):
            continue
        same_mask += 1
        d = np.linalg.norm(a - c, axis=-1)
        d = d[np.isfinite(d)]
        npk, nclose = npk + d.size, nclose + int((d <= 0.5).sum())
    return (f"peak-count diffs {[g - o for o, g in zip(npo, npg)]}; equal instance count {same_n}/{len(npo)}, equal assignment "
            f"{same_mask}/{len(npo)}, peaks within 0.5 px {nclose}/{npk}")


def run(name, cfg, w, frames, thr, scorer, stride, n_nodes, has_offsets):
    x = preprocess(frames)
    outs = {}
    for mode, kw in (("fp32", {}), ("bf16", dict(emulate_bf16=True, emulate_dtype=torch.bfloat16)), ("fp16", dict(emulate_bf16=True, emulate_dtype=torch.float16))):
        o = KerasGraph(cfg, w, **kw)(x)
        outs[mode] = (o[0], o[1], o[2] if has_offsets else None)
    print(f"== {name} (threshold {thr})")
    ref = post(*outs["fp32"], thr, scorer, stride)
    for mode in ("bf16", "fp16"):
        e = [float(np.abs(a - b).max() / np.abs(b).max()) for a, b in zip(outs[mode][:2], outs["fp32"][:2])]
        print(f"  {mode}: cms err {e[0]:.5f} pafs err {e[1]:.5f} of range; {compare(ref, post(*outs[mode], thr, scorer, stride), n_nodes)}")


def conditioning(cfg, w, frames, scorer, stride, n_nodes, thresholds=(0.5, 0.8), eps_list=(1e-5, 1e-4, 3e-4, 1e-3, 3e-3), seeds=3):
    cms, pafs, offs = KerasGraph(cfg, w)(preprocess(frames))[:3]
    for thr in thresholds:
        ref = post(cms, pafs, offs, thr, scorer, stride)
        print(f"== conditioning, threshold {thr}: fp32 oracle vs itself with maps + U(-eps, eps) x range")
        for eps in eps_list:
            for seed in range(seeds):
                rng = np.random.default_rng(seed)
                c2 = cms + (rng.random(cms.shape, dtype=np.float32) * 2 - 1) * np.float32(eps * np.abs(cms).max())
                p2 = pafs + (rng.random(pafs.shape, dtype=np.float32) * 2 - 1) * np.float32(eps * np.abs(pafs).max())
                print(f"  eps {eps:g} seed {seed}: {compare(ref, post(c2, p2, offs, thr, scorer, stride), n_nodes)}")


if __name__ == "__main__":
    from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights
    from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES

    cfg, w = load_npz_model(os.path.join("tests", "golden", "models", "minimal_instance.UNet.bottomup", "best_model.npz"))
    frames = render_frames(6, 384, 384, n_animals=2, seed=11)[0]
    sc = opg.PAFScorer(["A", "B"], [("A", "B")], 4, oob="zero")
    for thr in (0.2, 0.5, 0.8):
        run("trained fixture, synthetic frames", cfg, w, frames, thr, sc, 2, 2, True)
    conditioning(cfg, w, frames, sc, 2, 2)
