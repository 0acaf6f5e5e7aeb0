"""Is the network forward launch-bound at small batches, and what does a hipGraph replay of the same launch sequence buy?

    python tools/graph_probe.py [size]

Eager `DeviceNetwork.forward` (one ctypes call per launch from Python) vs `torch.cuda.CUDAGraph` capture of the same calls
(the kernels are enqueued on torch's current stream, all buffers are pre-allocated per shape, so the sequence is capturable).
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sleap_amd.benchmark_model import build_benchmark_predictor
from sleap_amd.synth import render_frames

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("| frames | eager ms | graph ms | outputs bitwise equal |\n|---|---|---|---|")
for B in (1, 2, 4, 8, 16, 64):
    pred, _, _ = build_benchmark_predictor(S, S, batch_size=B, seed=0)
    net = pred.inference_model.bottomup_layer.keras_model
    x = torch.from_numpy(render_frames(B, S, S, n_animals=4, seed=5)[0]).cuda()
    ref = [o.clone() for o in net.forward(x)]
    eager = timeit(lambda: net.forward(x), 50)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        net.forward(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        outs = net.forward(x)
    for o in outs:
        o.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(ref, outs))
    graph = timeit(g.replay, 50)
    print(f"| {B} | {eager:.3f} | {graph:.3f} | {same} |", flush=True)
