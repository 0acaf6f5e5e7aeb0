"""Host-side stages either side of the path, timed on the CPU -- no GPU involved.

    python tools/host_bench.py [n_frames]

  tracker   native `sa_tracker_track_frames` (host C++ in the kernel library) vs the Python restatement in oracle/tracking.py
            (which mirrors the reference's per-frame Python objects) on the same random-walk instances
  writer    predictions -> `.slp` tables (`io.slp.build_tables`, vectorised NumPy) and Labels views
  feed      FramePrefetcher over a memory-mapped .npy video: frames/s delivered as page-locked batches
The oracle import is for the timing comparison only (this is a tool, not a product path).
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import tracking as OT
from sleap_amd.io import slp
from sleap_amd.io.video import FramePrefetcher, Video
from sleap_amd.nn.tracking import Tracker

F = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
I, N = 4, 13
rng = np.random.default_rng(0)
base = rng.uniform(100, 900, (I, 1, 2)) + rng.normal(0, 20, (I, N, 2))
pts = np.empty((F, I, N, 2), np.float32)
for f in range(F):
    base = base + rng.normal(0, 2.0, (I, 1, 2))
    pts[f] = base + rng.normal(0, 0.5, (I, N, 2))
pts[rng.random((F, I, N)) < 0.03] = np.nan
vals = rng.uniform(0.2, 1, (F, I, N)).astype(np.float32)
scores = rng.uniform(0.3, 1, (F, I)).astype(np.float32)
n_valid = np.full((F,), I, np.int32)

print(f"{F} frames x {I} instances x {N} nodes")
for kw in (dict(tracker="simple", similarity="instance", match="greedy"),
           dict(tracker="simple", similarity="iou", match="hungarian"),
           dict(tracker="simplemaxtracks", similarity="object_keypoint", match="hungarian", max_tracks=4, max_tracking=True)):
    tr = Tracker.make_tracker_by_name(**kw)
    t0 = time.perf_counter()
    r = tr.track_frames(pts, vals, scores, n_valid, img_hw=(1024, 1024))
    dt = time.perf_counter() - t0
    n_o = min(F, 400)
    ot = OT.Tracker(**kw)
    t0 = time.perf_counter()
    for f in range(n_o):
        ot.track([OT.Inst(pts[f, i], vals[f, i], scores[f, i], uid=i) for i in range(I)], img_hw=(1024, 1024))
    do = (time.perf_counter() - t0) / n_o * F
    print(f"tracker {kw['tracker']}/{kw['similarity']}/{kw['match']}: native {F / dt:,.0f} frames/s ({dt * 1e3:.0f} ms), "
          f"Python restatement {F / do:,.0f} frames/s (extrapolated from {n_o} frames): x{do / dt:.0f}; "
          f"{len(np.unique(r['track'][r['track'] >= 0]))} tracks")

B = 64
outs = [dict(instance_peaks=pts[i:i + B], instance_peak_vals=vals[i:i + B], instance_scores=scores[i:i + B], n_valid=n_valid[i:i + B],
             frame_ind=np.arange(i, min(i + B, F)), video_ind=np.zeros(min(B, F - i), np.int64)) for i in range(0, F, B)]
t0 = time.perf_counter()
tables = slp.build_tables(outs)
dt = time.perf_counter() - t0
print(f"writer: build_tables {F / dt:,.0f} frames/s ({dt * 1e3:.0f} ms; {len(tables['instances'])} instances, {len(tables['pred_points'])} points)")

with tempfile.TemporaryDirectory() as d:
    T, H = 512, 1024
    path = os.path.join(d, "v.npy")
    np.save(path, rng.integers(0, 256, (T, H, H, 1), dtype=np.uint8))
    v = Video.from_filename(path)
    for rep in range(2):  # second pass: page cache warm
        t0 = time.perf_counter()
        n = 0
        for lo, hi, inds, batch in FramePrefetcher(v, [(i, min(i + B, T)) for i in range(0, T, B)], depth=4, pin_memory=False):
            n += hi - lo
        dt = time.perf_counter() - t0
        print(f"feed: memory-mapped .npy, {n} frames of {H}x{H} in {dt * 1e3:.0f} ms = {n / dt:,.0f} frames/s (pass {rep + 1})")
