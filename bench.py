#!/usr/bin/env python
"""Benchmark of the bottom-up inference hot path on MI355X (BASELINE.json metric:
"frames/sec at 1024x1024 bottom-up (13 nodes)").

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3]): bottom-up UNet + PAFs, 1024x1024x1 uint8 frames, 13 nodes / 12
edges (flies13), 4 animals per frame; architecture of training profile baseline_medium_rf.bottomup with
weights fitted to the synthetic fly video (sleap_amd/benchmark_model.py, tools/train_benchmark_model.py):
every frame yields 4 instances x 13 nodes, so matching and grouping run on real counts. `--batch` frames
per GPU per step (default 64; 64 UNIQUE rendered frames), frame-sharded over the GPUs (weak scaling),
results gathered with one all-gather. `--global-batch G` instead fixes the GLOBAL batch (strong scaling:
G / world frames per GPU -- configs[3] read literally is G = 64 over 8 GPUs).

A step = preprocessing (u8 -> float fused into the first conv) -> UNet forward -> local peaks with
integral refinement -> PAF scoring -> Hungarian matching -> instance assembly -> packed fixed-shape
results (all-gathered over ranks when N > 1), on frames already resident in HBM.

`python bench.py --gpus N` launches its N ranks itself (torch.distributed.run, one process per GPU, 127.0.0.1 rendezvous) and
refuses to run when fewer than N GPUs are visible or the process group ends up with a different rank count; under torchrun it
joins the group it was started in.

Rank 0 prints ONE JSON line (contract in the task statement): `value` = EXACTLY K timed steps between two barrier + synchronize
fences, MAX over ranks. Measured after that region, reported beside it: `sustained` (the same step repeated for >= 6 s),
`literal_split_8_per_gpu` (8 frames per GPU per step: configs[3]'s global batch of 64 over 8 GPUs), and two extra objects:
  roofline      conv3x3 MFMA kernel family: algorithmic TFLOP/s (2*H*W*Cin*Cout*9 over its launches) over
                its HIP-event-measured time inside this process, vs the 2.5 PFLOP/s dense fp16 / bf16 MFMA peak
  roofline_postproc  SURVEY 8(d)'s HBM fraction of the post-processing: the NMS scan's bytes (the confidence maps, once) over its
                HIP-event time, and the same bytes over the whole post-processing
  cpu_baseline  the CPU oracle (torch-CPU fp32 convs + NumPy/SciPy post-processing; "port") timed on the
                host cores over a bounded sample of the same workload (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense, fp16 and bf16 alike (/opt/skills/guides/MI355X_MICROARCH.md)
# HBM bytes of the conv kernel family per step of the DEFAULT workload (64 frames of 1024x1024) are NOT measured by this process
# (PMC counters need a rocprofv3 run of their own): `roofline.traffic` is read at run time from the newest
# profiles/r*_pmc_hbm_traffic.json (written by tools/pmc_traffic.py from two separate --pmc passes, FETCH_SIZE doubled per the
# gfx950 correction, WRITE_SIZE uncalibrated) and labelled with that file in `roofline.traffic_source`; null without one.


def profiled_traffic(batch, size):
    """-> (HBM bytes per step of the conv family, source label) from the newest tracked PMC summary matching this workload."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("frames_per_step") == batch and d.get("size") == size:
            return float(d["conv_family_bytes_per_step"]), f"profile file {os.path.relpath(f, ROOT)} (not measured by this run)"
    return None, "no PMC summary for this workload under profiles/"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fix the GLOBAL batch instead (strong scaling): frames per GPU = global batch / world size")
    ap.add_argument("--random-init", action="store_true",
                    help="round-1 stand-in model: seeded random weights with calibrated heads (no instances to group)")
    ap.add_argument("--parity-frames", type=int, default=0,
                    help="frames of the CPU-baseline / parity sample (0: as many as the time budget allows, at most one batch)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table to stderr")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the gather / barrier / max-reduce path even with one rank "
                         "(exercises the N > 1 code on a 1-GPU box)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="launcher / process-group check without a GPU: the ranks meet over gloo, gather a packed result block "
                         "and print the line's distributed fields (tests/test_bench_launcher.py)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the blocks measured after the timed region (sustained run, strong-scaling block, dense-activation pass)")
    ap.add_argument("--sustained-seconds", type=float, default=6.0,
                    help="wall-clock length of the `sustained` block after the timed region (longer than an SMI sampling period)")
    ap.add_argument("--dtype", choices=("bf16", "fp16"), default=None,
                    help="16-bit storage type of activations / conv weights (one library build each; accumulation is fp32). "
                         "Default: the package default (fp16, SLEAP_AMD_DTYPE)")
    return ap.parse_args()


def cpu_budget(limit=32):
    """CPUs this process may actually use: the cgroup quota when there is one (cpu.max), else the visible count, <= limit."""
    cores = min(os.cpu_count() or 1, limit)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(mc, weights, scorer_args, frames_u8, budget_s, device_result=None):
    """Times the CPU oracle on a bounded sample of the same frames (all host cores for the convs) and, as the checker,
    compares its instances with the device's for those frames (north_star: peaks within 0.5 px, identical assignments)."""
    from oracle import paf_grouping as opg
    from oracle import peak_finding as opf
    from oracle.keras_graph import KerasGraph, preprocess

    # The GPU box's container has a CPU quota (cgroup cpu.max: 16 CPUs of the 256 the OS reports): more threads than that
    # only spin and get the whole process throttled -- measured 0.22 s/frame at 16-32 threads, 0.40 s at 64, 0.73 s at 128,
    # 17.5 s at 256 (tests/diagnostics/cpu_probe.py). Use the quota when it can be read, at most 32 threads otherwise.
    cores = cpu_budget()
    torch.set_num_threads(cores)
    g = KerasGraph(mc, weights)
    scorer = opg.PAFScorer(scorer_args["nodes"], scorer_args["edges"], scorer_args["stride"], oob="zero")

    def run(batch):
        cms, pafs = g(preprocess(batch))[:2]
        pts, vals, si, ci = opf.find_local_peaks(cms, 0.2, "integral", 5)
        pts = pts * np.float32(4)
        B = batch.shape[0]
        return scorer.predict(pafs, [pts[si == b] for b in range(B)], [vals[si == b] for b in range(B)],
                              [ci[si == b] for b in range(B)])

    t0 = time.time()
    run(frames_u8[:1])  # warm-up (thread pools, allocator)
    t1 = time.time()
    per = max(t1 - t0, 1e-3)
    n = int(max(1, min(len(frames_u8), budget_s / per)))
    t2 = time.time()
    ref = run(frames_u8[:n])
    dt = time.time() - t2
    out = {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{n} frame(s) of the same 1024x1024 workload, one batch, after a 1-frame warm-up; "
                     f"torch-CPU fp32 convs ({cores} threads) + NumPy/SciPy post-processing"}
    if device_result is not None:
        # fp32 CPU network + reference post-processing vs the 16-bit-storage device path on the same frames, POSITIONALLY
        # (same n_valid per frame, same NaN mask, max ||delta(x, y)|| <= 0.5 px over non-NaN entries)
        max_d, bad_count, bad_mask, n_inst, n_pk, n_close, sum_d = 0.0, 0, 0, 0, 0, 0, 0.0
        max_dv, max_ds = 0.0, 0.0
        for f in range(n):
            want = np.asarray(ref[0][f], dtype=np.float32).reshape(-1, len(scorer_args["nodes"]), 2)
            nv = int(device_result["n_valid"][f])
            got = device_result["instance_peaks"][f, :nv].numpy()
            n_inst += len(want)
            if len(want) != nv:
                bad_count += 1
                continue
            if not np.array_equal(np.isnan(want), np.isnan(got)):
                bad_mask += 1
                continue
            if want.size:
                d = np.linalg.norm(got - want, axis=-1)
                ok = np.isfinite(d)
                if ok.any():
                    max_d = max(max_d, float(d[ok].max()))
                    n_pk += int(ok.sum())
                    n_close += int((d[ok] <= 0.5).sum())
                    sum_d += float(d[ok].sum())
                    max_dv = max(max_dv, float(np.nanmax(np.abs(device_result["instance_peak_vals"][f, :nv].numpy() - np.asarray(ref[1][f])))))
                    max_ds = max(max_ds, float(np.abs(device_result["instance_scores"][f, :nv].numpy() - np.asarray(ref[2][f])).max()))
        out["parity_vs_oracle"] = {"frames": n, "instances": n_inst, "peaks": n_pk, "peaks_within_0.5px": n_close,
                                   "max_peak_delta_px": round(max_d, 4), "mean_peak_delta_px": round(sum_d / max(n_pk, 1), 5),
                                   "max_peak_val_delta": round(max_dv, 5), "max_instance_score_delta": round(max_ds, 5),
                                   "frames_with_different_instance_count": bad_count,
                                   "frames_with_different_node_assignment": bad_mask,
                                   "tolerance_met": bool(bad_count == 0 and bad_mask == 0 and n_pk > 0 and n_close == n_pk)}
    return out


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks ourselves (one process per GPU, the command line the
    task statement gives for the driver) and hand their exit code back. No-op inside a launched rank or for N = 1."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    if not args.dry_run_cpu:
        n = torch.cuda.device_count()
        if n < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n} GPU(s) are visible -- refusing to report a "
                             f"{args.gpus}-GPU number from fewer devices")
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, cpu_budget() // args.gpus)))
    raise SystemExit(subprocess.call(cmd, env=env))


_JSON_FD = None


def claim_stdout():
    """stdout carries ONE JSON line (the driver parses it); libraries write there too -- RCCL prints a version banner at
    communicator creation, through C stdio, so it can land before or after Python's own output. From here on file descriptor 1
    is stderr for everything in this process; `emit` writes the line to the descriptor stdout had."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def dry_run_cpu(args, world, rank):
    """The distributed skeleton of a step on CPU tensors over gloo: shard arithmetic, one all-gather of the packed rows,
    barrier, max-reduce of the time -- everything `main` does around the GPU work. Prints the same distributed fields."""
    import torch.distributed as dist
    from sleap_amd import parallel

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} rank(s)")
    B = (args.global_batch // world) if args.global_batch else args.batch
    width = parallel.packed_width(32, 13)
    packed = torch.full((B, width), float(rank), dtype=torch.float32)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = parallel.gather_batch_results(packed, B * world, 32, 13, world)
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ok = bool(all(float(got[r * B, 0]) == float(r) for r in range(world)))
    # the second named curve of the N > 1 line (main: `configs3_global_batch_64`): 64 / N frames per rank through the same gather
    literal64 = None
    if not args.global_batch and 64 % world == 0 and B >= 64 // world:
        bl = 64 // world
        gl = parallel.gather_batch_results(packed[:bl].contiguous(), bl * world, 32, 13, world)
        literal64 = {"frames_per_gpu_per_step": bl, "global_batch": bl * world, "scaling": "strong",
                     "rows_in_rank_order": bool(all(float(gl[r * bl, 0]) == float(r) for r in range(world)))}
    if rank == 0:
        emit({"dry_run": True, "n_gpus": world, "steps": args.steps, "rows_in_rank_order": ok,
              "scaling": "strong" if args.global_batch else "weak",
              "value_weak_64_per_gpu": ("<value>" if (not args.global_batch and B == 64) else None),
              "value_strong_global_batch_64": ("<value>" if (args.global_batch and B * world == 64) else
                                               ("<configs3_global_batch_64.value>" if literal64 else None)),
              "configs3_global_batch_64": literal64,
              "config": {"n_ranks_seen": dist.get_world_size(), "collective_backend": dist.get_backend(),
                         "frames_per_gpu_per_step": B, "global_batch": B * world}})
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    self_launch(args)
    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch with `python bench.py --gpus N` "
                         "(self-launching) or torchrun --nproc-per-node N")
    if args.dry_run_cpu:
        return dry_run_cpu(args, world, rank)
    torch.cuda.set_device(local_rank)
    # torch's CPU kernels spin in OpenMP regions sized by the VISIBLE core count; under a container CPU quota that throttles
    # the whole process. The timed loop has no torch CPU ops, this keeps stray ones harmless.
    torch.set_num_threads(max(1, cpu_budget() // max(world, 1)))
    import torch.distributed as dist

    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} rank(s)")

    from sleap_amd import parallel
    from sleap_amd.benchmark_model import build_benchmark_predictor
    from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES, render_flies, render_frames

    from sleap_amd import _lib
    args.dtype = args.dtype or _lib.DEFAULT_DTYPE
    H = W = args.size
    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of the world size {world}")
        B = args.global_batch // world
    else:
        B = args.batch
    pred, mc, weights = build_benchmark_predictor(H, W, batch_size=B, seed=0, dtype=args.dtype, trained=not args.random_init)
    layer = pred.inference_model.bottomup_layer
    net = layer.keras_model
    scorer = layer.paf_scorer
    layer.assume_inputs_ready = True  # the frames are resident in HBM before the timed region (no producer to wait for)
    # B UNIQUE frames per rank (seeds < 10000; the model was fitted to seeds >= 10000), 4 animals each, resident in HBM
    render = render_frames if args.random_init else render_flies
    frames_np, _ = render(B, H, W, n_animals=4, seed=100 + rank)
    frames = torch.from_numpy(frames_np).cuda().contiguous()
    width = parallel.packed_width(scorer.max_instances, scorer.n_nodes)
    host_out = torch.empty((world * B, width), dtype=torch.float32).pin_memory()
    gathered = torch.empty((world * B, width), dtype=torch.float32, device="cuda")

    def step(fr=frames, g=gathered, h=host_out):
        outs = pred.inference_model.call(fr)
        packed = parallel.pack_results(outs)
        if use_dist:
            dist.all_gather_into_tensor(g, packed)
            h.copy_(g, non_blocking=True)
        else:
            h.copy_(packed, non_blocking=True)
        return outs

    def timed(n, **kw):
        """n steps between two fences -> seconds, the MAX over ranks."""
        fence()
        t = time.perf_counter()
        for _ in range(n):
            step(**kw)
        fence()
        t = time.perf_counter() - t
        if use_dist:
            tm = torch.tensor([t], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            t = float(tm.item())
        return t

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed pre-warm by TIME in front of the W warm-up steps: at small batches W = 3 steps are 2-3 ms, not enough for the chip
    # to leave its idle clocks (measured: the first 40 steps at 8 frames per step ran at 1.18 ms, the following thousand at 0.76)
    # (the step COUNT is agreed between the ranks first -- every step holds a collective)
    if world > 1:
        net.defer_range_agreement()  # kept three lines below, on every rank
    step()
    torch.cuda.synchronize()
    if world > 1:  # fp16 range scales: all ranks agree once, after their first forward (a no-op for a network that fits fp16)
        if net.dist_agree_range():
            step()
            torch.cuda.synchronize()
    t_pre = time.perf_counter()
    step()
    torch.cuda.synchronize()
    n_pre = int(min(max(0.3 / max(time.perf_counter() - t_pre, 1e-5), 1), 400))
    if use_dist:
        npt = torch.tensor([n_pre], dtype=torch.int64, device="cuda")
        dist.all_reduce(npt, op=dist.ReduceOp.MAX)
        n_pre = int(npt.item())
    for _ in range(n_pre):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    res = parallel.unpack_results(host_out[:B].clone(), scorer.max_instances, scorer.n_nodes)
    # digest of the gathered packed rows of the last timed step (rank 0's copy of the WHOLE job's results): the same frames give the
    # same digest with and without the process group (tests/test_gpu_dist_path.py compares the two)
    import hashlib

    result_digest = hashlib.sha1(np.ascontiguousarray(host_out.numpy()).tobytes()).hexdigest()[:16]
    status_bits = int(np.bitwise_or.reduce(res["status"].numpy().astype(np.int64)))
    mean_instances = float(res["n_valid"].float().mean())

    # ---- after the contract's K-step region (every rank takes part: the gather is in the step)
    sustained = small = literal64 = None
    if not args.no_extras:
        # (1) the same step repeated for >= --sustained-seconds of wall clock (default 6 s: longer than the driver's 5 s SMI
        # cadence, so that at least one of its samples sees the GPU busy; round 3's 1.2 s block fell between two samples)
        n_sus = max(args.steps, int(np.ceil(args.sustained_seconds * args.steps / max(dt, 1e-6))))
        if use_dist:  # (every step holds a collective: the count must be the same on every rank)
            nst = torch.tensor([n_sus], dtype=torch.int64, device="cuda")
            dist.all_reduce(nst, op=dist.ReduceOp.MAX)
            n_sus = int(nst.item())
        t_sus = timed(n_sus)
        sustained = {"steps_effective": n_sus, "seconds": round(t_sus, 3), "value": round(world * B * n_sus / t_sus, 2),
                     "ms_per_step": round(t_sus / n_sus * 1e3, 3)}
        # (2) configs[3] read literally -- a global batch of 64 frame-sharded over the job's GPUs. Two blocks, both named:
        #     `literal_split_8_per_gpu`: 8 frames per GPU per step, whatever N is (at N = 1 the projection of the 8-GPU point:
        #     its ms_per_step is what one of eight GPUs needs for its share of the 64-frame batch);
        #     `configs3_global_batch_64`: 64 / N frames per GPU per step on THIS run's N ranks -- the strong-scaling curve of the
        #     configuration BASELINE configs[3] names (N = 1: the timed region itself; N = 8: the same shape as the first block).
        def small_run(b_small):
            gs = torch.empty((world * b_small, width), dtype=torch.float32, device="cuda")
            hs = torch.empty((world * b_small, width), dtype=torch.float32).pin_memory()
            kw = dict(fr=frames[:b_small].contiguous(), g=gs, h=hs)
            timed(3, **kw)
            ns = max(args.steps, 50)
            ts = timed(ns, **kw)
            return {"frames_per_gpu_per_step": b_small, "global_batch": b_small * world, "steps": ns,
                    "value": round(world * b_small * ns / ts, 2), "ms_per_step": round(ts / ns * 1e3, 3)}

        b8 = 8
        if B > b8:
            small = small_run(b8)
        if not strong and 64 % world == 0 and B >= 64 // world:
            bl = 64 // world
            if bl == B:
                literal64 = {"frames_per_gpu_per_step": B, "global_batch": 64, "steps": args.steps,
                             "value": round(world * B * args.steps / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 3)}
            elif bl == b8 and small is not None:
                literal64 = dict(small)
            else:
                literal64 = small_run(bl)
            literal64["scaling"] = "strong"

    out = None
    if rank == 0:
        # ---- instrumented pass: per-op HIP events on the launch stream (outside the timed region)
        descs = net.op_descriptions(H, W)
        acc = np.zeros(len(descs))
        reps = 4
        for r in range(reps + 1):  # the first pass warms the per-launch Python loop (not counted); mean over the other four
            prof = []
            net.forward(layer.preprocess(frames), profile=prof)
            torch.cuda.synchronize()
            if r:
                acc += np.array([a.elapsed_time(b) for a, b in prof])
        acc /= reps  # ms per launch
        conv_ms = sum(ms for (k, _, _), ms in zip(descs, acc) if k == "conv")
        conv_fl = sum(f for (k, _, f) in descs if k == "conv") * B
        n_conv = sum(1 for (k, _, _) in descs if k == "conv")
        all_ms = float(acc.sum())
        # the same instrumented pass on DENSE activations (seeded random-init weights: the fitted network's post-ReLU maps are
        # sparse, the chip draws less power per MFMA and clocks higher -- the fitted-model fraction is data dependent)
        dense = None
        if not args.no_extras and not args.random_init and world == 1:
            from sleap_amd.benchmark_model import build_benchmark_graph
            from sleap_amd.nn.engine import DeviceNetwork

            _, mc_r, w_r = build_benchmark_graph(H, W, seed=0)
            net_r = DeviceNetwork(mc_r, w_r, dtype=args.dtype)
            accr = np.zeros(len(net_r.op_descriptions(H, W)))
            for r in range(3):
                prof = []
                net_r.forward(layer.preprocess(frames), profile=prof)
                torch.cuda.synchronize()
                if r:
                    accr += np.array([a.elapsed_time(b) for a, b in prof])
            accr /= 2
            descs_r = net_r.op_descriptions(H, W)
            dense = {"conv_ms": float(sum(ms for (k, _, _), ms in zip(descs_r, accr) if k == "conv")), "all_ms": float(accr.sum())}
            del net_r
        # the same plan with EVERY bilinear upsampling materialised by its own launch (the plan of rounds 1-2). Since round 3 the
        # default plan expands the upsampled source of the last decoder convolution inside that convolution (no launch, no HBM
        # tensor): its launch is then longer for the same FLOPs, so `frac` (conv launches only) of the default plan is not
        # comparable with earlier rounds' -- `frac_materialised` is, and `frac_forward` (all launches) compares either way.
        mat = None
        if not args.no_extras and world == 1 and any(nm.endswith("mode2") for _, nm, _ in descs):
            from sleap_amd.nn.engine import DeviceNetwork

            net_m = DeviceNetwork(mc, weights, dtype=args.dtype, fuse_upsample=False)
            descs_m = net_m.op_descriptions(H, W)
            accm = np.zeros(len(descs_m))
            for _ in range(3):  # (its buffers are new: a few un-instrumented passes first)
                net_m.forward(layer.preprocess(frames))
            for r in range(reps + 1):  # same protocol as the default plan's pass above
                prof = []
                net_m.forward(layer.preprocess(frames), profile=prof)
                torch.cuda.synchronize()
                if r:
                    accm += np.array([a.elapsed_time(b) for a, b in prof])
            accm /= reps
            mat = {"conv_ms": float(sum(ms for (k, _, _), ms in zip(descs_m, accm) if k == "conv")), "all_ms": float(accm.sum())}
            del net_m
        # post-processing time
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cms, pafs, offs = layer.forward_pass(frames)
        refinement = layer.refinement if layer.refinement in ("integral", "local") else None
        for rep in range(2):  # (the first call on this stream allocates and zeroes its scratch: not part of a steady-state step)
            torch.cuda.synchronize()
            e0.record()
            pp = scorer.predict_from_maps(cms, offs, pafs, layer.peak_threshold, refinement, layer.integral_patch_size,
                                          layer.cm_output_stride, layer.max_peaks)
            e1.record()
        torch.cuda.synchronize()
        post_ms = e0.elapsed_time(e1)
        mean_peaks = float(pp["peak_count"].float().mean())
        # achieved_hbm(postproc) = 3.41e6 B (the confidence maps of a frame: 256 x 256 x 13 f32, read once by the NMS
        # scan) x frames / time / 8 TB/s -- for the scan kernel alone (sa_find_local_peaks_rough, timed here on this stream: the
        # one pass of the post-processing that touches every map value) and over the WHOLE post-processing (scan + sort /
        # refinement + PAF scoring + matching + grouping: `post_ms` above; the part behind the scan is latency-bound, one
        # workgroup per frame)
        from sleap_amd import ops as _ops

        # (one launch per measurement, behind a 512-MiB fill: the maps of 64 frames are 218 MB and would otherwise be re-read from
        #  the 256-MiB Infinity Cache, not from HBM -- ten back-to-back launches measured 4.0 TB/s that way)
        evict = torch.empty((512 << 20,), dtype=torch.uint8, device="cuda")
        scans = []
        for rep in range(7):
            evict.fill_(rep)
            e0.record()
            _ops.find_local_peaks_rough(cms, layer.peak_threshold, layer.max_peaks)
            e1.record()
            torch.cuda.synchronize()
            scans.append(e0.elapsed_time(e1))
        del evict
        scan_ms = float(np.median(scans[1:]))
        cms_bytes = float(cms.numel() * cms.element_size())
        PEAK_HBM_GBS = 8000.0
        roofline_postproc = {
            "kernel": "nms_scan_kernel (find_local_peaks_rough: the confidence maps read once)", "bound": "hbm",
            "achieved": round(cms_bytes / (scan_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(cms_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "avg_launch_ms": round(scan_ms, 4),
            "algorithmic_bytes_per_frame": cms_bytes / B, "frames_per_launch": B,
            "whole_postproc": {"ms_per_step": round(post_ms, 3), "achieved": round(cms_bytes / (post_ms * 1e-3) / 1e9, 1),
                               "frac": round(cms_bytes / (post_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                               "note": "same bytes over scan + sort/refine + PAF scoring + matching + grouping; runs on the second stream "
                                       "under the next batch's network in the timed step"},
        }
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12
        roofline = {
            "kernel": f"conv3x3_dma_kernel + convpair / convpair64 + stem16_gray_kernel (the {n_conv} MFMA conv launches of one step)", "bound": "mfma", "achieved": round(achieved, 2),
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            # the same FLOPs over the TIMED step of the contract's K-step region (network + peak finding + PAF scoring + matching +
            # grouping + packing + gather + D2H, un-instrumented): the fraction the driver's clock implies. It can exceed
            # `frac_forward`, whose pass carries two HIP events per launch.
            "frac_step": round(conv_fl / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
            # the WHOLE network forward (the conv family + the launches without FLOPs: materialised upsampling) by the same FLOPs
            "frac_forward": round(conv_fl / (all_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "frac_dense": (round(conv_fl / (dense["conv_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if dense else None),
            "frac_forward_dense": (round(conv_fl / (dense["all_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if dense else None),
            "frac_materialised": (round(conv_fl / (mat["conv_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if mat else None),
            "frac_forward_materialised": (round(conv_fl / (mat["all_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if mat else None),
            "plan": ("default: the bilinear upsampling in front of the last decoder convolution is expanded in LDS inside that "
                     "convolution (per-layer rule of DeviceNetwork)" + ("; *_materialised = every upsampling as its own launch" if mat else "")
                     if any(nm.endswith("mode2") for _, nm, _ in descs) else "every upsampling materialised"),
            "traffic": profiled_traffic(B, H)[0] if H == W else None,
            "traffic_source": profiled_traffic(B, H)[1] if H == W else None,
            "traffic_unit": "HBM bytes per step over the kernel family's launches (FETCH_SIZE x 2 + WRITE_SIZE)",
            "launches_per_step": n_conv, "avg_launch_ms": round(conv_ms / n_conv, 4),
            "algorithmic_gflop_per_frame": round(conv_fl / B / 1e9, 2),
            "network_ms_per_step": round(all_ms, 3), "postproc_ms_per_step": round(post_ms, 3),
        }
        if args.layers:
            for (k, nm, f), ms in zip(descs, acc):
                tf = f * B / (ms * 1e-3) / 1e12 if ms > 0 else 0
                print(f"{nm:44s} {ms:8.3f} ms {tf:8.1f} TFLOP/s", file=sys.stderr)
        fps = world * B * args.steps / dt
        model_desc = ("random-init weights with calibrated heads" if args.random_init else
                      "weights fitted to the synthetic fly video (tools/train_benchmark_model.py)")
        out = {
            "metric": "frames/sec at 1024x1024 bottom-up (13 nodes)", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: bottom-up UNet(baseline_medium_rf: f16 r2 s32->4 bilinear)"
                                   f"+PAFs, {H}x{W}x1 u8, 13 nodes/12 edges, 4 animals, {model_desc}"
                                   + (f"; global batch {B * world} sharded over {world} GPU(s)" if strong else ""),
                       "frames_per_gpu_per_step": B, "global_batch": B * world, "unique_frames_per_gpu": int(frames.shape[0]),
                       "parallelism": f"frame-sharded dp{world}, one all-gather of packed results per step",
                       "n_ranks_seen": (dist.get_world_size() if use_dist else 1),
                       "collective_backend": (dist.get_backend() + " (RCCL)" if use_dist else None),
                       "activation_layout": "planes16" if net.planar else "nhwc",
                       "peak_threshold": 0.2, "refinement": "integral", "mean_peaks_per_frame": round(mean_peaks, 1),
                       "mean_instances_per_frame": round(mean_instances, 2), "status_bits": status_bits,
                       "result_digest": result_digest},
            "roofline": roofline, "roofline_postproc": roofline_postproc,
            "sustained": sustained, "literal_split_8_per_gpu": small,
            # the two readings of "batch = 64, frame-sharded over N GPUs", both named so that a SCALE record cannot be read as the
            # wrong one: `value` above is `value_weak_64_per_gpu` unless --global-batch was given (then it is the strong one)
            "value_weak_64_per_gpu": (round(fps, 2) if (not strong and B == 64) else None),
            "value_strong_global_batch_64": (round(fps, 2) if (strong and B * world == 64) else
                                             (literal64["value"] if literal64 else None)),
            "configs3_global_batch_64": literal64,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = frames_np if args.parity_frames <= 0 else frames_np[: args.parity_frames]
            out["cpu_baseline"] = cpu_baseline(mc, weights, {"nodes": FLIES13_NODES, "edges": FLIES13_EDGES, "stride": 8},
                                               sample, args.cpu_baseline_seconds, device_result=res)
        else:
            # the CPU oracle is timed on rank 0 at N = 1 only (task statement); an explicit block instead of null so that a
            # parser does not read the N > 1 line as "unmeasured"
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": None, "kind": "port",
                                   "reason": ("N>1: the CPU baseline is measured by the N=1 run only" if world > 1
                                              else "--no-cpu-baseline"),
                                   "sample": None}
        emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
