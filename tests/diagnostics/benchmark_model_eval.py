"""How well conditioned is the trained benchmark model (tools/train_benchmark_model.py) on fresh `render_flies` frames?
Runs the fp32 CPU oracle end to end and reports what the configs[3] parity tests depend on: instances per frame, distance to
the rendered ground truth, the margin of every detected peak above the 0.2 threshold and of every spurious local maximum below
it, and the margin of matched / unmatched PAF line scores around the 0.25 cut.

    python tests/diagnostics/benchmark_model_eval.py [weights.npz] [n_frames] [size]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import paf_grouping as opg  # noqa: E402
from oracle import peak_finding as opf  # noqa: E402
from oracle.keras_graph import KerasGraph, load_npz_model, preprocess  # noqa: E402
from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES, render_flies  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sleap_amd", "data", "benchmark_unet_flies13.npz")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    cfg, w = load_npz_model(path)
    w = {k: v.astype(np.float32) for k, v in w.items()}
    g = KerasGraph(cfg, w)
    frames, insts = render_flies(n, size, size, 4, seed=seed)
    t0 = time.time()
    cms, pafs = g(preprocess(frames))[:2]
    print(f"forward {time.time() - t0:.1f}s  cms {cms.shape} [{cms.min():.3f}, {cms.max():.3f}]  pafs {pafs.shape}")
    # every local maximum, thresholded at a tiny value, to see what sits near 0.2
    pts, vals, si, ci = opf.find_local_peaks(cms, 0.02, "integral", 5)
    print("local maxima > 0.02:", len(vals), " in (0.1, 0.3):", int(((vals > 0.1) & (vals < 0.3)).sum()),
          " > 0.2:", int((vals > 0.2).sum()), " expected:", sum(len(i) for i in insts) * 13)
    hi = vals[vals > 0.2]
    lo = vals[vals <= 0.2]
    print(f"detected peaks: min {hi.min():.3f} mean {hi.mean():.3f};  strongest sub-threshold maximum {lo.max() if len(lo) else 0:.3f}")
    pts, vals, si, ci = opf.find_local_peaks(cms, 0.2, "integral", 5)
    pts = pts * np.float32(4)
    scorer = opg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8, oob="zero")
    B = len(frames)
    out = scorer.predict(pafs, [pts[si == b] for b in range(B)], [vals[si == b] for b in range(B)], [ci[si == b] for b in range(B)])
    ninst = [len(x) for x in out[0]]
    print("instances per frame:", ninst)
    errs = []
    for b in range(B):
        pred = np.asarray(out[0][b]).reshape(-1, 13, 2)
        for gt in insts[b]:
            d = np.linalg.norm(pred - gt[None], axis=-1)  # (I, 13)
            j = int(np.nanargmin(np.nanmean(d, axis=1))) if len(pred) else -1
            if j >= 0:
                errs.append(d[j])
    errs = np.array(errs)
    print(f"node error vs rendered truth: mean {np.nanmean(errs):.2f} px, max {np.nanmax(errs):.2f} px, missing nodes {int(np.isnan(errs).sum())} of {errs.size}")
    ls = np.concatenate([np.asarray(x).reshape(-1) for x in out[5]])
    print(f"line scores: {len(ls)} candidates; > 0.25: {int((ls > 0.25).sum())}; in (0.1, 0.4): {int(((ls > 0.1) & (ls < 0.4)).sum())}; "
          f"matched-range min {ls[ls > 0.25].min() if (ls > 0.25).any() else float('nan'):.3f}")
    # the margins that decide the ASSIGNMENTS: the matched connections vs the 0.25 cut and vs the candidates left unmatched
    m = opg.match_candidates_batch(out[3], out[4], out[5], len(FLIES13_EDGES))
    ms = np.concatenate([np.asarray(x).reshape(-1) for x in m[3]])
    print(f"matched connections: {len(ms)} (expected {len(frames) * 4 * len(FLIES13_EDGES)}); min score {ms.min():.3f}; "
          f"matched but below 0.25: {int((ms < 0.25).sum())}")
    unmatched = []
    for b in range(B):
        ls, left = np.asarray(out[5][b]).tolist(), np.asarray(m[3][b]).tolist()
        for v in left:
            ls.remove(v)
        unmatched += ls
    print(f"strongest candidate the matching did NOT choose: {max(unmatched):.3f} (weakest chosen one: {ms.min():.3f})")
    sc = np.concatenate([np.asarray(x).reshape(-1) for x in out[2]])
    print("instance scores:", np.round(sc, 2))


if __name__ == "__main__":
    main()
