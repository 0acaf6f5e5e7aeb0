"""Per-launch timing of a backbone built by sleap_amd.nn.architectures on random weights (HIP events between plan ops).

    python tools/net_profile.py resnet50|hourglass|unet [H] [B] [min_share]

Every launch is listed (round 5: the tracked ResNet-50 table had dropped the launches below 1.2 % of the total -- 29 of 58 --
and the other 1.76 ms only appeared in the by-kind sums); `min_share` > 0 restores a filter for quick looks.
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from sleap_amd.nn import architectures as A
from sleap_amd.nn.engine import DeviceNetwork

kind = sys.argv[1]
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
if kind == "resnet50":  # BASELINE configs[4]: ResNet-50 encoder + upsampling stack, 24 nodes / 23 edges
    cfg, sh = A.build_resnet_model_config((H, H, 1), "ResNet50", 32, pretrained=True,
                                          upsampling=dict(output_stride=4, method="transposed_conv", skip_connections="concatenate",
                                                          filters=64, refine_convs=2),
                                          heads=[("MultiInstanceConfmapsHead", 24, 4), ("PartAffinityFieldsHead", 46, 8)])
elif kind == "hourglass":
    cfg, sh = A.build_hourglass_model_config((H, H, 1), stem_stride=4, max_stride=64, output_stride=4, stem_filters=128,
                                             filters=256, filter_increase=128, stacks=1,
                                             heads=[("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 4)])
else:
    cfg, sh = A.build_unet_model_config((H, H, 1), 16, 2, 32, 4, True, True,
                                        heads=[("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 8)])
net = DeviceNetwork(cfg, A.he_normal_weights(sh, 0, residual_scale=0.25))
x = torch.randint(0, 256, (B, H, H, 1), dtype=torch.uint8, device="cuda")
for _ in range(2):
    net.forward(x)
torch.cuda.synchronize()
descs = net.op_descriptions(H, H)
nbytes = net.op_bytes(H, H)  # algorithmic HBM bytes per frame and launch (every input read once, every output written once)
assert len(nbytes) == len(descs)
acc = np.zeros(len(descs))
reps = 3
for _ in range(reps):
    prof = []
    net.forward(x, profile=prof)
    torch.cuda.synchronize()
    acc += np.array([a.elapsed_time(b) for a, b in prof])
acc /= reps
agg = {}
for (k, nm, f), ms, nb in zip(descs, acc, nbytes):
    key = nm.split(" @")[0].split(" ")[0] if k != "conv" else nm.split(" ")[0]
    a = agg.setdefault(key, [0.0, 0.0, 0, 0.0])
    a[0] += ms
    a[1] += f * B
    a[2] += 1
    a[3] += nb * B
    if ms >= float(sys.argv[4] if len(sys.argv) > 4 else 0.0) * acc.sum():
        print(f"{nm:52s} {ms:8.3f} ms {f * B / ms / 1e9 if ms else 0:8.1f} TFLOP/s {nb * B / ms / 1e9 if ms else 0:7.2f} TB/s  "
              f"({f / max(nb, 1):5.0f} FLOP/B)")
print("---- by kind (TB/s = algorithmic bytes: every input read once, every output written once)")
for key, (ms, fl, n, nb) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{key:20s} n={n:3d} {ms:8.3f} ms {fl / ms / 1e9 if ms else 0:8.1f} TFLOP/s {nb / ms / 1e9 if ms else 0:7.2f} TB/s")
tot_f = sum(f for _, _, f in descs) * B
tot_b = float(sum(nbytes)) * B
print(f"total {acc.sum():.3f} ms for {B} frames of {H}x{H}: {B / acc.sum() * 1e3:.1f} frames/s, {tot_f / acc.sum() / 1e9:.1f} TFLOP/s, "
      f"{tot_f / B / 1e9:.1f} GFLOP/frame")
# both rooflines of the whole forward (round 6, VERDICT r5 item 8: for a backbone of 1x1 convs the MFMA fraction is the wrong
# yardstick -- its launches move their bytes at 3-4.5 TB/s): FLOPs / time / 2.5 PFLOP/s, and ALGORITHMIC bytes (every launch's
# inputs read once, outputs written once) / time / 8 TB/s; the share of the time spent in launches that move > 3 TB/s
hbm_ms = sum(ms for ms, nb in zip(acc, nbytes) if ms > 0 and nb * B / ms / 1e9 > 3.0)
print(f"roofline of the forward: MFMA {tot_f / acc.sum() / 1e9 / 2500.0:.3f} of 2.5 PFLOP/s; HBM {tot_b / acc.sum() / 1e9 / 8.0:.3f} of 8 TB/s "
      f"({tot_b / 1e9:.2f} GB algorithmic per {B} frames, {tot_b / acc.sum() / 1e9:.2f} TB/s average); "
      f"{hbm_ms / acc.sum():.0%} of the time in launches above 3 TB/s")
