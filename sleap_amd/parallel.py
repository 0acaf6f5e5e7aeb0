"""Frame-sharded data parallelism for the bottom-up path.

The reference has no multi-GPU code. Here every global batch of consecutive frames is split into
contiguous per-rank ranges (frame order is preserved by concatenation); weights are replicated; each
rank packs its fixed-shape results into ONE float32 buffer and a single all-gather (RCCL over xGMI
when the backend is "nccl", gloo on CPU for tests) assembles the batch. The payload is
B/world * (I*N*3 + I + 2) floats -- tens of KB, latency-bound, never chunked.
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(i0: int, i1: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice of frames [i0, i1) owned by `rank`: ceil-sized shards, trailing ranks may be empty."""
    n = i1 - i0
    per = (n + world - 1) // world
    lo = min(i0 + rank * per, i1)
    hi = min(lo + per, i1)
    return lo, hi


def packed_width(max_instances: int, n_nodes: int) -> int:
    return max_instances * n_nodes * 3 + max_instances + 2


def pack_results(outs: Dict[str, torch.Tensor]) -> torch.Tensor:
    """instance_peaks | instance_peak_vals | instance_scores | n_valid | status -> (b, width) float32."""
    b = outs["instance_peaks"].shape[0]
    return torch.cat([
        outs["instance_peaks"].reshape(b, -1),
        outs["instance_peak_vals"].reshape(b, -1),
        outs["instance_scores"].reshape(b, -1),
        outs["n_valid"].to(torch.float32).reshape(b, 1),
        outs["status"].to(torch.float32).reshape(b, 1),
    ], dim=1).contiguous()


def unpack_results(packed: torch.Tensor, max_instances: int, n_nodes: int) -> Dict[str, torch.Tensor]:
    b = packed.shape[0]
    I, N = max_instances, n_nodes
    o = 0
    peaks = packed[:, o : o + I * N * 2].reshape(b, I, N, 2)
    o += I * N * 2
    vals = packed[:, o : o + I * N].reshape(b, I, N)
    o += I * N
    scores = packed[:, o : o + I]
    o += I
    return {"instance_peaks": peaks, "instance_peak_vals": vals, "instance_scores": scores,
            "n_valid": packed[:, o].to(torch.int32), "status": packed[:, o + 1].to(torch.int32)}


def unpack_results_np(packed, max_instances: int, n_nodes: int):
    """`unpack_results` for a host NumPy array. Deliberately NumPy: torch CPU ops enter OpenMP parallel regions whose idle
    threads spin, and under a container CPU quota (cpu.max 16 CPUs of 256 on the GPU boxes) a few of them per batch exhaust
    the quota -- the whole process then stalls for the rest of the 100 ms CFS period (measured: 88 ms stalls every third
    batch in the predict loop, tools/predict_e2e.py)."""
    b = packed.shape[0]
    I, N = max_instances, n_nodes
    o = I * N * 2
    return {"instance_peaks": packed[:, :o].reshape(b, I, N, 2), "instance_peak_vals": packed[:, o:o + I * N].reshape(b, I, N),
            "instance_scores": packed[:, o + I * N:o + I * N + I], "n_valid": packed[:, o + I * N + I].astype(np.int32),
            "status": packed[:, o + I * N + I + 1].astype(np.int32)}


def gather_batch_results(packed: Optional[torch.Tensor], n_batch: int, max_instances: int, n_nodes: int, world: int,
                         device=None) -> torch.Tensor:
    """All ranks contribute their (b_r, width) block (possibly empty); returns the (n_batch, width) batch.

    One collective per batch: blocks are padded to the common per-rank size so a single
    `all_gather_into_tensor` suffices."""
    if world == 1:
        return packed
    width = packed_width(max_instances, n_nodes)
    per = (n_batch + world - 1) // world
    if device is None:
        device = packed.device if packed is not None else torch.device("cpu")
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    send = torch.full((per, width), float("nan"), dtype=torch.float32, device=device)
    if packed is not None and packed.shape[0] > 0:
        send[: packed.shape[0]] = packed.to(device)
    recv = torch.empty((world * per, width), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(recv, send)
    # shards are contiguous and ceil-sized, so the first n_batch rows of the rank-major buffer are the batch
    return recv[:n_batch]
