"""What does the per-step result gather cost? One rank, RCCL process group of size 1 (the collective degenerates to a copy, but
the launch path -- ProcessGroupNCCL, its stream and events, the watchdog thread -- is the real one).

    python tools/dist_probe.py
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from sleap_amd import parallel
from sleap_amd.benchmark_model import build_benchmark_predictor
from sleap_amd.synth import render_frames

B, S = 64, 1024
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
pred, _, _ = build_benchmark_predictor(S, S, batch_size=B, seed=0)
layer = pred.inference_model.bottomup_layer
layer.assume_inputs_ready = True
if len(sys.argv) > 1 and sys.argv[1] == "hi":
    layer._net_stream = torch.cuda.Stream(priority=-1)
    print("network stream: high priority")
sc = layer.paf_scorer
frames = torch.from_numpy(render_frames(8, S, S, n_animals=4, seed=100)[0]).cuda().repeat(8, 1, 1, 1).contiguous()
width = parallel.packed_width(sc.max_instances, sc.n_nodes)
host_out = torch.empty((B, width), dtype=torch.float32).pin_memory()
gathered = torch.empty((B, width), dtype=torch.float32, device="cuda")
comm = torch.cuda.Stream()
pending = []


def step(mode):
    outs = pred.inference_model.call(frames)
    packed = parallel.pack_results(outs)
    if mode == "none":
        host_out.copy_(packed, non_blocking=True)
    elif mode == "copy":
        gathered.copy_(packed)
        host_out.copy_(gathered, non_blocking=True)
    elif mode == "sync":
        dist.all_gather_into_tensor(gathered, packed)
        host_out.copy_(gathered, non_blocking=True)
    elif mode == "async":
        w = dist.all_gather_into_tensor(gathered, packed, async_op=True)
        pending.append(w)
        if len(pending) > 1:
            pending.pop(0).wait()
            host_out.copy_(gathered, non_blocking=True)
    elif mode == "side-deferred":  # the gather of step k is queued after the network of step k+1
        if pending:
            ev, pk = pending.pop(0)
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                dist.all_gather_into_tensor(gathered, pk)
                host_out.copy_(gathered, non_blocking=True)
            pk.record_stream(comm)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((ev, packed))
    elif mode == "side":
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(comm):
            comm.wait_event(ev)
            dist.all_gather_into_tensor(gathered, packed)
            host_out.copy_(gathered, non_blocking=True)
        packed.record_stream(comm)


def run(mode, n=30):
    for _ in range(5):
        step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{mode:8s} {dt * 1e3:.3f} ms/step  {B / dt:.0f} frames/s", flush=True)


run("none")
run("copy")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
run("none")
run("copy")
run("sync")
run("async")
pending.clear()
run("side")
run("side-deferred")
pending.clear()
run("none")
dist.destroy_process_group()
