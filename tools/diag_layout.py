"""Diagnostic: run a network as 16-channel planes and as NHWC and report, per plan buffer, where the stored tensors differ.
    python tools/diag_layout.py hourglass|unet [H=128] [W=160]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from sleap_amd.nn import architectures as A
from sleap_amd.nn.engine import DeviceNetwork

kind = sys.argv[1]
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
W = int(sys.argv[3]) if len(sys.argv) > 3 else 160
heads = [("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 4)]
if kind == "hourglass":
    cfg, sh = A.build_hourglass_model_config((H, W, 1), stem_stride=4, max_stride=32, output_stride=4, stem_filters=32, filters=64,
                                             filter_increase=32, stacks=1, heads=heads)
else:
    cfg, sh = A.build_unet_model_config((H, W, 1), 16, 2, 32, 4, True, True, heads=heads)
w = A.he_normal_weights(sh, 1, residual_scale=0.25)
x = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (3, H, W, 1), dtype=np.uint8)).cuda()
kw = dict(fuse_heads=False) if len(sys.argv) > 4 else {}
a, b = DeviceNetwork(cfg, w, **kw), DeviceNetwork(cfg, w, layout="nhwc", **kw)
print("planar:", a.planar, b.planar, "plans equal:", [o[0] for o in a.plan] == [o[0] for o in b.plan])
oa, ob = a.forward(x), b.forward(x)
torch.cuda.synchronize()
descs = a.op_descriptions(H, W)
for (op, d) in zip(a.plan, descs):
    outs = [t for t in DeviceNetwork._writes(op, None)] if False else []
for i in sorted(a.buf_meta):
    ta, tb = a.stored_tensor(i).float(), b.stored_tensor(i).float()
    dd = (ta - tb).abs()
    n = int((dd > 0).sum())
    where = ""
    if n:
        idx = torch.nonzero(dd > 0)
        lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
        where = f" first {idx[0].tolist()} bbox {lo}..{hi}"
    print(f"buf {i:3d} {tuple(ta.shape)} max|a| {float(ta.abs().max()):9.3f} differing {n:8d} max diff {float(dd.max()):.4g}{where}")
for k, (p, q) in enumerate(zip(oa, ob)):
    print("output", k, tuple(p.shape), "max diff", float((p - q).abs().max()))
for k, (op, d) in enumerate(zip(a.plan, descs)):
    print(k, d[1])
