"""Probe: does splitting the batch over S streams (S independent sub-batches in flight) hide the launch tails of the network?
    python tools/two_stream_probe.py [B=64] [H=1024] [reps=30]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sleap_amd.benchmark_model import build_benchmark_predictor
from sleap_amd.synth import render_flies

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
frames_np, _ = render_flies(B, H, H, n_animals=4, seed=100)
frames = torch.from_numpy(frames_np).cuda().contiguous()


def nets(n):
    out = []
    for _ in range(n):
        pred, _, _ = build_benchmark_predictor(H, H, batch_size=B // n, seed=0, trained=True)
        out.append(pred.inference_model.bottomup_layer.keras_model)
    return out


for S in (1, 2, 4, 1, 2, 4):
    ns = nets(S)
    streams = [torch.cuda.Stream() for _ in range(S)]
    parts = [frames[i * (B // S):(i + 1) * (B // S)].contiguous() for i in range(S)]

    def run():
        for n, st, fr in zip(ns, streams, parts):
            with torch.cuda.stream(st):
                n.forward(fr)

    for _ in range(40):
        run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / reps
    print(f"streams={S} sub-batch={B // S}: {t * 1e3:.3f} ms per {B} frames = {B / t:.0f} frames/s (network only)", flush=True)
    del ns
