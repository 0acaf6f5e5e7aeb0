"""The reference's default call on its default input format, end to end: an .mp4 (H.264) -> `MediaVideo` (the package's decoder,
GOPs decoded ahead on host threads) -> `Predictor.predict` of a SLEAP-trained model folder -> instances, and optionally the flow
tracker. Prints frames/s of the whole call and how many frames hold two instances (the video shows two flies).

    python tools/predict_mp4.py [video.mp4] [model dir | benchmark] [tracker]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mp4 = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "video", "centered_pair_low_quality.mp4")
model = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "models", "minimal_instance.UNet.bottomup")
from sleap_amd.io.video import Video, VideoReader
from sleap_amd.nn.inference import load_model

if model == "benchmark":  # the bench line's fitted flies13 UNet (1024 x 1024, 13 nodes): the headline network on a REAL 1024 x 1024 video
    from sleap_amd.benchmark_model import build_benchmark_predictor

    predictor = build_benchmark_predictor(1024, 1024, batch_size=64, seed=0)[0]
else:
    predictor = load_model(model, batch_size=64, progress_reporting="none")
if len(sys.argv) > 3:
    from sleap_amd.nn.tracking import Tracker

    predictor.tracker = Tracker.make_tracker_by_name(tracker=sys.argv[3], track_window=5)
video = Video.from_filename(mp4)
print(f"{mp4}: {video.shape}, key frames {video.backend.keyframes[:4]}..., decode workers {video.backend._gops.workers if video.backend._gops else 1}")
t0 = time.perf_counter()
frames = video.get_frames(list(range(len(video))))
t_dec = time.perf_counter() - t0
print(f"decode alone: {len(video)} frames in {t_dec:.2f} s = {len(video) / t_dec:.0f} frames/s")
predictor.predict(frames[:128], make_labels=False)  # warm-up
for rep in range(2):
    video = Video.from_filename(mp4)  # a fresh backend: nothing decoded yet
    t0 = time.perf_counter()
    outs = predictor.predict(VideoReader(video), make_labels=False)
    dt = time.perf_counter() - t0
    nv = np.concatenate([o["n_valid"] for o in outs])
    print(f"predict(VideoReader(mp4)): {len(nv)} frames in {dt:.2f} s = {len(nv) / dt:.0f} frames/s; instances per frame: "
          f"{ {int(k): int((nv == k).sum()) for k in np.unique(nv)} }", flush=True)
    if predictor.tracker is not None:
        predictor.tracker.reset_candidates()
