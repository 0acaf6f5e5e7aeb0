import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from sleap_amd.nn import architectures as A
from sleap_amd.nn.engine import DeviceNetwork
cfg, sh = A.build_unet_model_config((256, 256, 1), 24, 2, 16, 4, True, True, heads=[("CenteredInstanceConfmapsHead", 13, 4)])
w = A.he_normal_weights(sh, 0)
x = torch.randint(0, 256, (164, 256, 256, 1), dtype=torch.uint8, device="cuda")
outs = {}
for name, kw in (("fused stem", {}), ("fuse_stem=False", dict(fuse_stem=False)), ("use_stem16=False", dict(use_stem16=False)), ("mfma_stem=False", dict(mfma_stem=False))):
    net = DeviceNetwork(cfg, w, **kw)
    for _ in range(3): o = net.forward(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): o = net.forward(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    outs[name] = o[0].clone()
    prof = []; net.forward(x, profile=prof); torch.cuda.synchronize()
    print(f"{name}: {dt*1e3:.3f} ms per 164 crops; planar={net.planar}; first launches (ms): " + " ".join(f"{a.elapsed_time(b):.3f}" for a, b in prof[:4]))
base = outs["fused stem"]
for k, v in outs.items(): print(k, float((v - base).abs().max()))
