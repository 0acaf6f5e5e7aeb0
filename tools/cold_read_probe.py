"""What a cold read of N MB reaches on this box (a reduction over a float32 buffer, behind a 512-MiB evicting fill, one launch per
measurement; median of 7): the yardstick for `roofline_postproc`'s scan (218 MB per 64 frames)."""
import torch

evict = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for mb in (54, 109, 218, 436, 872, 1744):
    x = torch.rand(mb * 1000 * 1000 // 4, dtype=torch.float32, device="cuda")
    for fn_name, fn in (("sum", lambda: x.sum()), ("amax", lambda: x.amax())):
        ts = []
        for _ in range(7):
            evict.fill_(1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        ms = ts[len(ts) // 2]
        print(f"{fn_name:5s} {mb:5d} MB cold: {ms * 1e3:7.1f} us = {mb / ms / 1e3:.2f} TB/s", flush=True)
