"""Rate of the flow tracker (device Lucas-Kanade candidates) on 1024x1024 frames: 4 animals x 13 nodes, track_window 5.
usage: python tools/flow_bench.py [n_frames]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sleap_amd import ops
from sleap_amd.nn.tracking import Tracker
from sleap_amd.synth import render_flies

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
# one rendered frame gliding by (2, 3) pixels per step: every point has a true correspondence, as in a video (frames that have
# nothing to do with each other make every point run its 30 steps on all 4 levels: the worst case, 0.31 ms per launch)
f0, inst = render_flies(1, 1024, 1024, n_animals=4, seed=3)
frames = np.stack([np.roll(f0[0], (3 * t, 2 * t), axis=(0, 1)) for t in range(F)])
pts = np.stack([np.asarray(inst[0], np.float32) + np.array([2 * t, 3 * t], np.float32) for t in range(F)])  # (F, 4, 13, 2)
nv = np.full((F,), pts.shape[1], np.int32)
dev = torch.from_numpy(frames).cuda()
print(f"| what | frames | ms / frame | frames / s |\n|---|---|---|---|")
for kw in (dict(tracker="flow"), dict(tracker="flowmaxtracks", max_tracks=4, max_tracking=True), dict(tracker="simple")):
    warm = Tracker.make_tracker_by_name(**kw)
    warm.track_frames(pts[:8], None, None, nv[:8], img_hw=(1024, 1024), images=dev[:8] if warm.uses_image else None)
    tr = Tracker.make_tracker_by_name(**kw)  # (reset_candidates keeps a max-tracks tracker's tracks with EMPTY queues, as the reference does)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.track_frames(pts, None, None, nv, img_hw=(1024, 1024), images=dev if tr.uses_image else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"| tracker {kw['tracker']} (window 5, 4 x 13 points, frames resident on the device) | {F} | {dt / F * 1e3:.3f} | {F / dt:,.0f} |")
p0, p1 = ops.FlowPyramid(dev[0]), ops.FlowPyramid(dev[1])
q = torch.from_numpy(pts[0].reshape(-1, 2)).cuda()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n_rep in (1, 5):
    pp = q.repeat(n_rep, 1)
    ops.optical_flow_pyr_lk(p0, p1, pp)
    e0.record()
    for _ in range(20):
        ops.optical_flow_pyr_lk(p0, p1, pp)
    e1.record()
    torch.cuda.synchronize()
    print(f"| sa_flow_lk, {pp.shape[0]} points, window 21, 4 levels | | {e0.elapsed_time(e1) / 20:.3f} ms / call | |")
e0.record()
for _ in range(20):
    ops.FlowPyramid(dev[0])
e1.record()
torch.cuda.synchronize()
print(f"| sa_flow_pyramid_build 1024x1024 (incl. allocation) | | {e0.elapsed_time(e1) / 20:.3f} ms / call | |")
