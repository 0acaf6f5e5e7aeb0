#!/usr/bin/env python
"""Fits the BENCHMARK model -- UNet `baseline_medium_rf.bottomup` (filters 16, rate 2, max_stride 32, output_stride 4, bilinear
upsampling; 13-node confidence maps @ stride 4 + 12-edge part-affinity fields @ stride 8) -- to the synthetic fly video of
`sleap_amd.synth.render_flies`, so that the benchmark (bench.py) and the configs[3] end-to-end parity tests run on a network
that DETECTS the rendered animals (4 instances x 13 nodes per frame, peaks far above the 0.2 threshold, PAF scores far above
the 0.25 cut) instead of on the noise-like maps of a random-init network.

TOOLING, not product: plain torch autograd on whatever device torch offers (CPU here; a few hundred steps on 256 x 256 crops
are enough because node identity and limb direction are locally decodable in the rendering). Seeded and reproducible on one
machine/torch build. The targets are the reference's own training targets for this profile, re-stated in torch:

    confidence maps   sleap/nn/data/confidence_maps.py:10-110   exp(-d^2 / 2 sigma^2), sigma = 2.5 px, max over instances,
                                                                grid = 0, 4, 8, ... image pixels (data/utils.py:41-70)
    PAFs              sleap/nn/data/edge_maps.py:16-211         gaussian_pdf(SQUARED distance to the segment, sigma = 75)
                                                                (the reference squares twice: exp(-d^4 / 2 sigma^2), i.e. a
                                                                ~9 px wide field) x unit vector, summed over instances

    python tools/train_benchmark_model.py --steps 1500 --out sleap_amd/data/benchmark_unet_flies13.npz

The stored weights are rounded to fp16-representable values (the fp32 oracle and the fp16-storage device path then hold
IDENTICAL weights; only activation rounding separates them) and saved as float16 arrays + the Keras-style graph JSON, the
same layout sleap_amd/nn/_h5_extract.py writes for a real `best_model.h5`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights  # noqa: E402  (plain data, no GPU)
from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES, render_flies  # noqa: E402

EDGE_IDX = [(FLIES13_NODES.index(a), FLIES13_NODES.index(b)) for a, b in FLIES13_EDGES]


class TorchGraph(torch.nn.Module):
    """Differentiable executor of the Keras functional-graph description (the layer types a bilinear UNet uses)."""

    def __init__(self, model_config, weights):
        super().__init__()
        cfg = model_config["config"]
        self.layers = cfg["layers"]
        self.output_names = [l[0] for l in cfg["output_layers"]]
        self.params = torch.nn.ParameterDict()
        self.keys = {}
        for k, v in weights.items():
            pk = k.replace("/", "__").replace(".", "_")
            self.keys[k] = pk
            self.params[pk] = torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))

    def w(self, k):
        return self.params[self.keys[k]]

    def forward(self, x):  # x: (B, 1, H, W) float in [0, 1]
        t = {}
        for l in self.layers:
            cn, name, c = l["class_name"], l["name"], l["config"]
            if cn == "InputLayer":
                t[name] = x
                continue
            ins = [t[n[0]] for n in l["inbound_nodes"][0]]
            if cn == "Conv2D":
                k = self.w(f"{name}/kernel").permute(3, 2, 0, 1)
                y = F.conv2d(ins[0], k, self.w(f"{name}/bias"), padding=k.shape[2] // 2)
                if c.get("activation", "linear") == "relu":
                    y = F.relu(y)
            elif cn == "Activation":
                y = F.relu(ins[0]) if c["activation"] == "relu" else ins[0]
            elif cn == "MaxPooling2D":
                y = F.max_pool2d(ins[0], 2, 2)
            elif cn == "UpSampling2D":
                y = F.interpolate(ins[0], scale_factor=2, mode="bilinear", align_corners=False)
            elif cn == "Concatenate":
                y = torch.cat(ins, dim=1)
            else:
                raise NotImplementedError(cn)
            t[name] = y
        return [t[n] for n in self.output_names]


def targets(insts, h, w, device, cm_sigma=2.5, paf_sigma=75.0):
    """insts: list (per sample) of (A, 13, 2) arrays in crop pixel coordinates -> cms (B,13,h/4,w/4), pafs (B,24,h/8,w/8)."""
    B = len(insts)
    xv4 = torch.arange(0, w, 4, device=device, dtype=torch.float32)
    yv4 = torch.arange(0, h, 4, device=device, dtype=torch.float32)
    xv8 = torch.arange(0, w, 8, device=device, dtype=torch.float32)
    yv8 = torch.arange(0, h, 8, device=device, dtype=torch.float32)
    cms = torch.zeros((B, 13, len(yv4), len(xv4)), device=device)
    pafs = torch.zeros((B, 24, len(yv8), len(xv8)), device=device)
    gx, gy = xv8[None, None, :], yv8[None, :, None]
    for b, inst in enumerate(insts):
        if len(inst) == 0:
            continue
        p = torch.as_tensor(np.asarray(inst), device=device, dtype=torch.float32)  # (A, 13, 2)
        dx = xv4[None, None, None, :] - p[:, :, 0, None, None]
        dy = yv4[None, None, :, None] - p[:, :, 1, None, None]
        cms[b] = torch.exp(-(dx * dx + dy * dy) / (2 * cm_sigma ** 2)).amax(dim=0)
        for e, (s, d) in enumerate(EDGE_IDX):
            src, dst = p[:, s], p[:, d]  # (A, 2)
            v = dst - src
            ln2 = (v * v).sum(-1).clamp(min=1.0)
            rx, ry = gx - src[:, 0, None, None], gy - src[:, 1, None, None]
            t_ = ((rx * v[:, 0, None, None] + ry * v[:, 1, None, None]) / ln2[:, None, None]).clamp(0, 1)
            d2 = (t_ * v[:, 0, None, None] - rx) ** 2 + (t_ * v[:, 1, None, None] - ry) ** 2
            em = torch.exp(-(d2 * d2) / (2 * paf_sigma ** 2))  # the reference's double squaring
            u = v / (v * v).sum(-1, keepdim=True).sqrt()
            pafs[b, 2 * e] = (em * u[:, 0, None, None]).sum(0)
            pafs[b, 2 * e + 1] = (em * u[:, 1, None, None]).sum(0)
    return cms, pafs


def sample_batch(rng, pool_frames, pool_insts, batch, crop):
    """Random crops (flip-free: left/right node identity is part of the task) around random animals of the frame pool."""
    T, H, W, _ = pool_frames.shape
    xs = np.empty((batch, 1, crop, crop), np.float32)
    insts = []
    for b in range(batch):
        t = rng.integers(T)
        a = pool_insts[t][rng.integers(len(pool_insts[t]))]
        c = a[1] + rng.uniform(-crop * 0.35, crop * 0.35, 2)  # thorax + offset
        x0 = int(np.clip(c[0] - crop / 2, 0, W - crop)) // 8 * 8
        y0 = int(np.clip(c[1] - crop / 2, 0, H - crop)) // 8 * 8
        xs[b, 0] = pool_frames[t, y0:y0 + crop, x0:x0 + crop, 0].astype(np.float32) * np.float32(1 / 255)
        insts.append(pool_insts[t] - np.array([x0, y0], np.float32))
    return xs, insts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--crop", type=int, default=256)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--pool", type=int, default=96, help="1024x1024 training frames rendered up front")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "sleap_amd", "data", "benchmark_unet_flies13.npz"))
    ap.add_argument("--resume", default=None)
    ap.add_argument("--margin", type=int, default=1, help="add the threshold-margin terms to the loss")
    ap.add_argument("--bg-weight", type=float, default=20.0, help="weight of the background-margin term")
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    torch.manual_seed(args.seed)
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")

    mc, shapes = build_unet_model_config((1024, 1024, 1), 16, 2.0, 32, 4, True, True, None,
                                         heads=[("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 8)])
    weights = he_normal_weights(shapes, seed=args.seed)
    if args.resume:
        z = np.load(args.resume)
        weights = {k: z[k].astype(np.float32) for k in z.files if k != "__model_config__"}
    net = TorchGraph(mc, weights).to(dev)
    print(f"{sum(p.numel() for p in net.parameters())} parameters on {dev}", flush=True)
    frames, insts = render_flies(args.pool, 1024, 1024, 4, seed=10_000 + args.seed + (1000 if args.resume else 0))  # training seeds are disjoint from bench/test seeds (< 10000)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=args.lr, total_steps=args.steps, pct_start=0.1)
    margin = 24  # crop borders see zero padding where the full frame has context: keep them out of the loss
    t0 = time.time()
    for step in range(args.steps):
        xs, ins = sample_batch(rng, frames, insts, args.batch, args.crop)
        x = torch.from_numpy(xs).to(dev)
        tc, tp = targets(ins, args.crop, args.crop, dev)
        cms, pafs = net(x)
        m4, m8 = margin // 4, margin // 8
        wc = 1.0 + 30.0 * tc  # sparse positives: weight them up
        lc = (wc * (cms - tc) ** 2)[:, :, m4:-m4, m4:-m4].mean()
        lp = ((1.0 + 5.0 * tp.abs()) * (pafs - tp) ** 2)[:, :, m8:-m8, m8:-m8].mean()
        # conditioning terms: what the parity tests need is not a small MSE but MARGINS -- no background response anywhere near
        # the 0.2 peak threshold, every true peak far above it (the target's own peak value varies between 0.53 and 1 with the
        # sub-grid position of the point at sigma 2.5 px / stride 4)
        bgm = (tc < 0.02).float()
        lm = args.bg_weight * ((F.relu(cms - 0.06) ** 2) * bgm)[:, :, m4:-m4, m4:-m4].mean()
        is_pk = (tc > 0.5) & (tc >= F.max_pool2d(tc, 3, 1, 1))
        lm = lm + 5.0 * ((F.relu(0.5 - cms) ** 2) * is_pk.float())[:, :, m4:-m4, m4:-m4].sum() / max(int(is_pk.sum()), 1)
        loss = lc + lp + (lm if args.margin else 0.0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)  # a single bad step at the top of the LR cycle once zeroed the PAF branch
        opt.step()
        sched.step()
        if step % 25 == 0 or step == args.steps - 1:
            with torch.no_grad():
                pk = (cms * (tc > 0.5)).amax(dim=(2, 3))
                has = (tc > 0.5).flatten(2).any(-1)
                bg = (cms * (tc < 0.01)).amax()
            print(f"step {step:5d} loss {loss.item():.5f} (cm {lc.item():.5f} paf {lp.item():.5f}) "
                  f"peak@gt mean {pk[has].mean().item():.3f} min {pk[has].min().item():.3f} max bg {bg.item():.3f} "
                  f"{time.time() - t0:.0f}s", flush=True)
        if (step + 1) % 250 == 0 or step == args.steps - 1:
            save(net, mc, weights, args.out)
    print("done", time.time() - t0, flush=True)


def save(net, mc, weights, out):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    d = {}
    for k in weights:
        d[k] = net.w(k).detach().cpu().numpy().astype(np.float16)  # fp16-representable: oracle and device share them exactly
    d["__model_config__"] = np.frombuffer(json.dumps(mc).encode("utf-8"), dtype=np.uint8)
    np.savez(out, **d)
    print("saved", out, os.path.getsize(out), flush=True)


if __name__ == "__main__":
    main()
