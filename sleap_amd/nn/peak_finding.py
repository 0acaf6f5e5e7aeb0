"""Peak finding with the reference's function surface (`sleap/nn/peak_finding.py`), executed by the
HIP kernels in csrc/postproc.hip.

Inputs may be NumPy arrays or torch tensors of shape (samples, height, width, channels); results
are torch CUDA tensors. Flat (ragged-concatenated) outputs are ordered exactly as the reference's:
row-major over (sample, y, x, channel).
"""
from typing import Optional, Tuple

import torch

from .. import _lib, ops


class PeakOverflowError(RuntimeError):
    """More local peaks in a frame than the fixed-shape buffer holds (raise `max_peaks`)."""


def _flatten(peak_xy, peak_val, peak_chan, peak_count, status, max_peaks):
    st = int(status.max().item()) if status.numel() else 0
    if st & _lib.STATUS_PEAK_OVERFLOW:
        raise PeakOverflowError(f"a frame has more than max_peaks={max_peaks} local peaks")
    B, P = peak_val.shape
    mask = torch.arange(P, device=peak_val.device)[None, :] < peak_count[:, None]
    sample_inds = torch.arange(B, device=peak_val.device, dtype=torch.int32)[:, None].expand(B, P)
    return peak_xy[mask], peak_val[mask], sample_inds[mask], peak_chan[mask]


def find_local_peaks_rough(cms, threshold: float = 0.2, max_peaks: int = 1024):
    """peak_finding.py:249-308. Returns (peak_points (n,2) [x,y], peak_vals, peak_sample_inds, peak_channel_inds)."""
    return find_local_peaks(cms, threshold=threshold, refinement=None, max_peaks=max_peaks)


def find_local_peaks(cms, threshold: float = 0.2, refinement: Optional[str] = None,
                     integral_patch_size: int = 5, max_peaks: int = 1024
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """peak_finding.py:451-532. `refinement` in {None, "integral", "local"}; anything else = None."""
    cms = ops.to_cuda_f32(cms)
    if refinement not in ("integral", "local"):
        refinement = None
    out = ops.find_local_peaks(cms, None, threshold, refinement, integral_patch_size, 1.0, max_peaks)
    return _flatten(*out, max_peaks)


def find_local_peaks_integral(cms, crop_size: int = 5, threshold: float = 0.2, max_peaks: int = 1024):
    """peak_finding.py:535-563."""
    return find_local_peaks(cms, threshold=threshold, refinement="integral", integral_patch_size=crop_size,
                            max_peaks=max_peaks)


def find_local_peaks_with_offsets(cms, offsets, threshold: float = 0.2, max_peaks: int = 1024):
    """peak_finding.py:646-707. `offsets` is (samples, height, width, 2 * channels)."""
    cms = ops.to_cuda_f32(cms)
    offsets = ops.to_cuda_f32(offsets)
    out = ops.find_local_peaks(cms, offsets, threshold, "offsets", 0, 1.0, max_peaks)
    return _flatten(*out, max_peaks)


def find_global_peaks_rough(cms, threshold: float = 0.1):
    """peak_finding.py:193-246. Returns (peak_points (samples, channels, 2), peak_vals (samples, channels))."""
    return find_global_peaks(cms, threshold=threshold, refinement=None)


def find_global_peaks(cms, threshold: float = 0.2, refinement: Optional[str] = None, integral_patch_size: int = 5):
    """peak_finding.py:337-420."""
    cms = ops.to_cuda_f32(cms)
    if refinement not in ("integral", "local"):
        refinement = None
    return ops.find_global_peaks(cms, None, threshold, refinement, integral_patch_size, 1.0)


def find_global_peaks_integral(cms, crop_size: int = 5, threshold: float = 0.2):
    """peak_finding.py:423-448."""
    return find_global_peaks(cms, threshold=threshold, refinement="integral", integral_patch_size=crop_size)


def find_global_peaks_with_offsets(cms, offsets, threshold: float = 0.2):
    """peak_finding.py:566-643."""
    cms = ops.to_cuda_f32(cms)
    offsets = ops.to_cuda_f32(offsets)
    return ops.find_global_peaks(cms, offsets, threshold, "offsets", 0, 1.0)
