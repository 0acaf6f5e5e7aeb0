import torch
x = torch.empty((64, 256, 256, 128), dtype=torch.bfloat16, device="cuda")
y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = x.numel() * 2 / 1e9
ms = t(lambda: x.fill_(1.0)); print(f"fill {gb:.2f} GB: {ms:.4f} ms {gb/ms:.2f} TB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy {gb:.2f} GB: {ms:.4f} ms {2*gb/ms:.2f} TB/s read+written")
ms = t(lambda: x.sum()); print(f"sum  {gb:.2f} GB: {ms:.4f} ms {gb/ms:.2f} TB/s read")
