"""Debug aid: per-layer max relative error of the HIP engine vs the (bf16-emulating) oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle.keras_graph import KerasGraph, ensure_float
from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights
from sleap_amd.nn.engine import DeviceNetwork

if len(sys.argv) > 1:
    from sleap_amd.nn.engine import load_keras_npz
    from sleap_amd.synth import render_frames
    cfg, w = load_keras_npz(sys.argv[1])
    x = render_frames(2, 128, 160, 2, seed=1)[0]
else:
    cfg, shapes = build_unet_model_config((128, 160, 1), 16, 2, 32, 4, True, True,
                                          heads=[("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 8)])
    w = he_normal_weights(shapes, seed=1)
    rng = np.random.default_rng(2)
    x = rng.integers(0, 256, (2, 128, 160, 1), dtype=np.uint8)
net = DeviceNetwork(cfg, w)
outs = net.forward(torch.from_numpy(x).cuda())
torch.cuda.synchronize()
bufs = net._buffers[(2, 128, 160)]
_, allt = KerasGraph(cfg, w, emulate_bf16=True)(ensure_float(x), return_all=True)
_, allf = KerasGraph(cfg, w, emulate_bf16=False)(ensure_float(x), return_all=True)
# map plan ops to layer names: conv ops are in layer order
conv_layers = [l["name"] for l in cfg["config"]["layers"] if l["class_name"] in ("Conv2D", "Conv2DTranspose")]
acts = {}
for l in cfg["config"]["layers"]:
    if l["class_name"] == "Activation":
        acts[l["inbound_nodes"][0][0][0]] = l["name"]
i = 0
for op in net.plan:
    if op[0] in ("stem", "conv", "head", "convt"):
        name = conv_layers[i]; i += 1
        o = op[1] if op[0] == "stem" else (op[6] if op[0] == "conv" else (op[4] if op[0] == "convt" else op[2]))
        if o.buf is None:
            continue
        ref_name = acts.get(name, name)
        d = bufs[o.buf].float().cpu().numpy()[..., :o.c]
        r, rf = allt[ref_name], allf[ref_name]
        e = np.abs(d - r).max() / np.abs(r).max()
        ef = np.abs(d - rf).max() / np.abs(rf).max()
        nbad = int((np.abs(d - r) > 1e-6 * np.abs(r).max()).sum())
        print(f"{op[0]:5s} {name:44s} c={o.c:3d} max|ref|={np.abs(r).max():9.3e} vs bf16-oracle {e:.3e} (n_diff {nbad}/{d.size})  vs fp32 {ef:.3e}")
