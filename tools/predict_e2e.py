"""End-to-end rate of the user-facing call, HOST frames in -> result dictionaries out (PCIe included):
`Predictor.predict(frames, make_labels=False)` on a (T, 1024, 1024, 1) uint8 NumPy array in pageable host memory, batch 64.
The FramePrefetcher stages this rank's batches in page-locked buffers and uploads them on the network stream while the previous
batch computes. Reported next to the HBM-resident bench line in DESIGN.md section 5 (it is never bench.py's `value`).

    python tools/predict_e2e.py [T] [labels|arrays] [tracker name, e.g. flow]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sleap_amd.benchmark_model import build_benchmark_predictor
from sleap_amd.synth import render_frames

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
labels = len(sys.argv) > 2 and sys.argv[2] == "labels"
import os
BATCH = int(os.environ.get("PRED_BATCH", "64"))
pred, _, _ = build_benchmark_predictor(1024, 1024, batch_size=BATCH, seed=0)
if len(sys.argv) > 3:
    from sleap_amd.nn.tracking import Tracker

    pred.tracker = Tracker.make_tracker_by_name(tracker=sys.argv[3], track_window=5)
base = render_frames(16, 1024, 1024, n_animals=4, seed=100)[0]
frames = np.ascontiguousarray(np.tile(base, (T // 16, 1, 1, 1)))
pred.predict(frames[:128], make_labels=False)  # warm-up: buffers, pinned ring
for rep in range(3):
    t0 = time.perf_counter()
    out = pred.predict(frames, make_labels=labels)
    dt = time.perf_counter() - t0
    n = len(out) if labels else sum(len(o["n_valid"]) for o in out)
    if pred.tracker is not None:
        pred.tracker.reset_candidates()
    print(f"predict(make_labels={labels}, tracker={sys.argv[3] if len(sys.argv) > 3 else None}): {T} host frames in {dt * 1e3:.1f} ms = {T / dt:.0f} frames/s ({n} results)", flush=True)
