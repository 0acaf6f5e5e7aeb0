#!/usr/bin/env python
"""Fits small networks of the architectures BASELINE.json configs[0], [1], [2] and [4] name to the synthetic videos of
`sleap_amd.synth.render_animals`, so that the end-to-end parity tests of those configurations (tests/test_gpu_config_parity.py)
run on networks that DETECT what is in the frames -- confident, well separated peaks -- as tools/train_benchmark_model.py
does for configs[3]. TOOLING, not product: plain torch autograd on the CPU; seeded.

    python tools/train_config_models.py c0_single5 c1_single13 c2_centroid c2_centered c4_resnet [--steps N] [--threads T]

    task          architecture (reference training profile)                        input                       head(s)
    c0_single5    UNet f16 r2 s16->2 bilinear (baseline.centroid.json)             256^2 x0.5, 1 animal        SingleInstanceConfmapsHead 5 @2
    c1_single13   UNet f16 r2 s16->2 bilinear (baseline_medium_rf.single.json)     512^2, 1 animal             SingleInstanceConfmapsHead 13 @2
    c2_centroid   UNet f16 r2 s16->2 bilinear (baseline.centroid.json)             1024^2 x0.5, 2 animals      CentroidConfmapsHead 1 @2
    c2_centered   UNet f24 r2 s16->4 bilinear (baseline_medium_rf.topdown.json)    160^2 crops                 CenteredInstanceConfmapsHead 13 @4
    c4_resnet     ResNet-50 (resnet.py:544-595, imagenet preprocessing Lambdas) +  1024^2, 8 animals           MultiInstanceConfmapsHead 24 @4,
                  UpsamplingStack (transposed conv k4 s2 + BN, concatenate skips)                              PartAffinityFieldsHead 46 @8

Targets are the reference's own (sleap/nn/data/confidence_maps.py:10-110, edge_maps.py:16-211, instance_centroids.py), as in
tools/train_benchmark_model.py. Weights are stored as **float32 masters** (real SLEAP weights are fp32: the device path rounds
them to its 16-bit storage type itself, the fp32 oracle does not). For `c4_resnet` only the stem, conv2, conv3, the upsampling
stack and the heads are fitted and stored (~5 M parameters); conv4 / conv5 (22 M parameters) keep their seeded He-normal
values, which `sleap_amd.nn.architectures.he_normal_weights(shapes, seed=0)` regenerates bit for bit (a checksum of them is
stored and verified at load), with BatchNormalization statistics calibrated on the video and stored.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # config_models.py: test infrastructure

from sleap_amd.nn import architectures as A  # noqa: E402  (plain data, no GPU)
from sleap_amd import synth  # noqa: E402

from config_models import ANCHOR, DATA_DIR as OUT_DIR, TASKS, load_task_weights, task_graph  # noqa: E402


class TorchGraph(torch.nn.Module):
    """Differentiable executor of the Keras functional-graph description (the layer set of SURVEY.md 8a). BatchNormalization
    runs on batch statistics while `self.bn_batch_stats` is set (updating the moving statistics with momentum 0.9), on the
    moving statistics otherwise (what inference does)."""

    def __init__(self, model_config, weights, freeze=None):
        super().__init__()
        cfg = model_config["config"]
        self.layers = cfg["layers"]
        self.output_names = [l[0] for l in cfg["output_layers"]]
        self.params = torch.nn.ParameterDict()
        self.keys = {}
        self.bn_batch_stats = True
        self.frozen = set()
        for k, v in weights.items():
            pk = k.replace("/", "__").replace(".", "_")
            self.keys[k] = pk
            t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
            stat = k.endswith(("/moving_mean", "/moving_variance"))
            fr = bool(freeze and re.search(freeze, k))
            if stat or fr:
                self.register_buffer(pk, t)
                if fr and not stat:
                    self.frozen.add(k)
            else:
                self.params[pk] = torch.nn.Parameter(t)

    def w(self, k):
        pk = self.keys[k]
        return self.params[pk] if pk in self.params else getattr(self, pk)

    def forward(self, x):  # x: (B, C, H, W) float in [0, 1]
        t = {}
        for l in self.layers:
            cn, name, c = l["class_name"], l["name"], l["config"]
            if cn == "InputLayer":
                t[name] = x
                continue
            ins = [t[n[0]] for n in l["inbound_nodes"][0]]
            if cn == "Conv2D":
                k = self.w(f"{name}/kernel").permute(3, 2, 0, 1)
                s, d = c["strides"][0], c.get("dilation_rate", [1, 1])[0]
                xin = ins[0]
                if c["padding"] == "same":
                    keff = (k.shape[2] - 1) * d + 1
                    n = xin.shape[2]
                    tot = max((-(-n // s) - 1) * s + keff - n, 0)
                    xin = F.pad(xin, (tot // 2, tot - tot // 2, tot // 2, tot - tot // 2))
                y = F.conv2d(xin, k, self.w(f"{name}/bias"), stride=s, dilation=d)
                if c.get("activation", "linear") == "relu":
                    y = F.relu(y)
            elif cn == "Conv2DTranspose":
                k = self.w(f"{name}/kernel").permute(3, 2, 0, 1)
                s = c["strides"][0]
                y = F.conv_transpose2d(ins[0], k, None, stride=s)
                ch = max(k.shape[2] - s, 0) // 2
                y = y[:, :, ch:ch + ins[0].shape[2] * s, ch:ch + ins[0].shape[3] * s] + self.w(f"{name}/bias").view(1, -1, 1, 1)
                if c.get("activation", "linear") == "relu":
                    y = F.relu(y)
            elif cn == "BatchNormalization":
                mm, mv = self.w(f"{name}/moving_mean"), self.w(f"{name}/moving_variance")
                y = F.batch_norm(ins[0], mm, mv, self.w(f"{name}/gamma"), self.w(f"{name}/beta"),
                                 training=self.bn_batch_stats, momentum=0.1, eps=c.get("epsilon", 1e-3))
            elif cn == "Activation":
                y = F.relu(ins[0]) if c["activation"] == "relu" else ins[0]
            elif cn == "MaxPooling2D":
                y = F.max_pool2d(ins[0], c["pool_size"][0], c["strides"][0])
            elif cn == "ZeroPadding2D":
                (pt, pb), (pl, pr) = c["padding"]
                y = F.pad(ins[0], (pl, pr, pt, pb))
            elif cn == "UpSampling2D":
                y = F.interpolate(ins[0], scale_factor=2, mode=c.get("interpolation", "nearest"),
                                  **({"align_corners": False} if c.get("interpolation") == "bilinear" else {}))
            elif cn == "Concatenate":
                y = torch.cat(ins, dim=1)
            elif cn == "Add":
                y = ins[0] + ins[1]
            elif cn == "Lambda":
                if name == "tile_channels":
                    y = ins[0].repeat(1, 3, 1, 1)
                elif name == "imagenet_preproc_v1":
                    y = (ins[0] * 255.0).flip(1) - torch.tensor([103.939, 116.779, 123.68], device=ins[0].device).view(1, 3, 1, 1)
                else:
                    raise NotImplementedError(name)
            else:
                raise NotImplementedError(cn)
            t[name] = y
        return [t[n] for n in self.output_names]


def cm_targets(points, h, w, stride, sigma=2.5):
    """points: (B, C, K, 2) with NaN for absent -> (B, C, h/stride, w/stride): max over K of exp(-d^2 / 2 sigma^2)."""
    xv = torch.arange(0, w, stride, dtype=torch.float32)
    yv = torch.arange(0, h, stride, dtype=torch.float32)
    p = torch.as_tensor(points, dtype=torch.float32)
    dx = xv[None, None, None, None, :] - p[..., 0, None, None]
    dy = yv[None, None, None, :, None] - p[..., 1, None, None]
    g = torch.exp(-(dx * dx + dy * dy) / (2 * sigma ** 2))
    return torch.nan_to_num(g, nan=0.0).amax(dim=2)


def paf_targets(insts, edge_idx, h, w, stride=8, sigma=75.0):
    """insts: list (per sample) of (A, N, 2) -> (B, 2E, h/stride, w/stride), the reference's double-squared distance field."""
    xv = torch.arange(0, w, stride, dtype=torch.float32)
    yv = torch.arange(0, h, stride, dtype=torch.float32)
    gx, gy = xv[None, None, :], yv[None, :, None]
    out = torch.zeros((len(insts), 2 * len(edge_idx), len(yv), len(xv)))
    for b, inst in enumerate(insts):
        if len(inst) == 0:
            continue
        p = torch.as_tensor(np.asarray(inst), dtype=torch.float32)
        for e, (s, d) in enumerate(edge_idx):
            src, dst = p[:, s], p[:, d]
            v = dst - src
            ln2 = (v * v).sum(-1).clamp(min=1.0)
            rx, ry = gx - src[:, 0, None, None], gy - src[:, 1, None, None]
            t_ = ((rx * v[:, 0, None, None] + ry * v[:, 1, None, None]) / ln2[:, None, None]).clamp(0, 1)
            d2 = (t_ * v[:, 0, None, None] - rx) ** 2 + (t_ * v[:, 1, None, None] - ry) ** 2
            em = torch.exp(-(d2 * d2) / (2 * sigma ** 2))
            u = v / (v * v).sum(-1, keepdim=True).sqrt()
            out[b, 2 * e] = (em * u[:, 0, None, None]).sum(0)
            out[b, 2 * e + 1] = (em * u[:, 1, None, None]).sum(0)
    return out


def scale_frames(frames_u8, scale):
    """ensure_float then resize_image (resizing.py:71-105: bilinear, half-pixel centres, no antialias) -> (T, 1, h, w) float."""
    x = torch.from_numpy(frames_u8[..., 0].astype(np.float32) * np.float32(1 / 255))[:, None]
    if scale != 1.0:
        x = F.interpolate(x, size=(int(x.shape[2] * scale), int(x.shape[3] * scale)), mode="bilinear", align_corners=False)
    return x


def scale_points(p, scale):
    return p if scale == 1.0 else (p + 0.5) * scale - 0.5


def sample(rng, t, pool_x, pool_insts, batch):
    """-> x (B, 1, c, c), per-sample target description."""
    T, _, H, W = pool_x.shape
    crop = t["crop"] or H
    xs = torch.empty((batch, 1, crop, crop))
    insts = []
    for b in range(batch):
        f = rng.integers(T)
        a = pool_insts[f][rng.integers(len(pool_insts[f]))]
        if t["kind"] == "centered":
            c = a[ANCHOR] + rng.uniform(-3, 3, 2)  # the crop is centred on the (predicted) centroid
            x0, y0 = int(round(c[0] - crop / 2)), int(round(c[1] - crop / 2))
        elif t["crop"] is None:
            x0 = y0 = 0
        else:
            c = a.mean(0) + rng.uniform(-crop * 0.3, crop * 0.3, 2)
            x0 = int(np.clip(c[0] - crop / 2, 0, W - crop))
            y0 = int(np.clip(c[1] - crop / 2, 0, H - crop))
        # zero-padded crop (frames near the border)
        xs[b] = 0
        sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x0 + crop, W), min(y0 + crop, H)
        xs[b, 0, sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = pool_x[f, 0, sy0:sy1, sx0:sx1]
        off = np.array([x0, y0], np.float32)
        insts.append((pool_insts[f] - off, a - off))
    return xs, insts


def make_targets(t, skel, insts, crop):
    kind = t["kind"]
    if kind in ("single", "centered"):
        pts = np.stack([a for _, a in insts])[:, :, None, :]  # (B, N, 1, 2): THIS animal's nodes only
        return [cm_targets(pts, crop, crop, t["heads"][0][2])]
    if kind == "centroid":
        A_ = max(len(al) for al, _ in insts)
        pts = np.full((len(insts), 1, A_, 2), np.nan, np.float32)
        for b, (al, _) in enumerate(insts):
            pts[b, 0, :len(al)] = al[:, ANCHOR]
        return [cm_targets(pts, crop, crop, t["heads"][0][2])]
    A_ = max(len(al) for al, _ in insts)
    N = len(skel.nodes)
    pts = np.full((len(insts), N, A_, 2), np.nan, np.float32)
    for b, (al, _) in enumerate(insts):
        pts[b, :, :len(al)] = al.transpose(1, 0, 2)
    return [cm_targets(pts, crop, crop, t["heads"][0][2]),
            paf_targets([al for al, _ in insts], skel.edge_idx, crop, crop, t["heads"][1][2])]


def fit(task, steps=None, threads=0, seed=0, lr=2e-3, resume=False, device="cpu", out_dir=None, batch=0, hard_neg=0.0,
        bn_inference=False, crop=0):
    """`device="cuda"`: the same fit on a GPU (plain torch; used for the ResNet-50 task, 45 minutes on 8 CPU cores). The stored
    weights then depend on the GPU's convolution algorithms: reproducible in distribution, not bit for bit."""
    t = dict(TASKS[task])
    steps = steps or t["steps"]
    if batch:
        t["batch"] = batch
    if crop:
        t["crop"] = crop
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    skel = getattr(synth, t["skeleton"])
    frames, insts = synth.render_animals(t["pool"], t["frame"], t["frame"], t["n_animals"], seed=10_000 + seed, skeleton=skel,
                                         margin=t["render_margin"], body=t.get("body", (80.0, 120.0)),
                                         min_sep=t.get("min_sep", 170.0))
    pool_x = scale_frames(frames, t["input_scale"])
    pool_insts = [scale_points(a, t["input_scale"]) for a in insts]
    crop = t["crop"] or pool_x.shape[2]
    mc, shapes = task_graph(t, crop, crop)
    out = os.path.join(out_dir or OUT_DIR, f"config_{task}.npz")
    if resume and os.path.exists(out):
        _, weights = load_task_weights(task, crop, crop, out, seed)
    else:
        weights = A.he_normal_weights(shapes, seed=seed)
        for k in weights:  # BatchNormalization starts neutral; the video sets the moving statistics
            if k.endswith("/gamma") or k.endswith("/moving_variance"):
                weights[k][:] = 1.0
            elif k.endswith("/beta") or k.endswith("/moving_mean"):
                weights[k][:] = 0.0
    net = TorchGraph(mc, weights, freeze=t.get("freeze")).to(device)
    n_fit = sum(p.numel() for p in net.parameters())
    print(f"[{task}] {n_fit} fitted parameters ({len(net.frozen)} frozen tensors), crop {crop}, {steps} steps", flush=True)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=steps, pct_start=0.1)
    # pixels of a crop's border ring left out of the loss. 0 for tasks whose network sees whole frames at inference with a large
    # receptive field (the ResNet: with a free border ring its confidence maps reached 4.0 along the frame edges -- 1100 false
    # peaks per 1024 x 1024 frame -- because a crop edge and a frame edge are the same zero padding to the network)
    margin = t.get("loss_margin", 16 if t["crop"] else 0)
    has_bn = any(l["class_name"] == "BatchNormalization" for l in net.layers)
    t0 = time.time()
    if bn_inference:
        net.bn_batch_stats = False  # a resumed fit: the moving statistics are calibrated, fit the inference-time function only
    for step in range(steps):
        if has_bn and step == int(steps * 0.7):
            net.bn_batch_stats = False  # the last 30 %: the inference-time function (moving statistics) is what gets fitted
        xs, ins = sample(rng, t, pool_x, pool_insts, t["batch"])
        tg = [y.to(device) for y in make_targets(t, skel, ins, crop)]
        outs = net(xs.to(device))
        loss, parts = 0.0, []
        for i, (o, y) in enumerate(zip(outs, tg)):
            m = margin // t["heads"][i][2]
            sl = (slice(None), slice(None), slice(m, o.shape[2] - m), slice(m, o.shape[3] - m))
            if i == 0:
                l = ((1.0 + 30.0 * y) * (o - y) ** 2)[sl].mean()
                bg = (y < 0.02).float()
                l = l + 20.0 * ((F.relu(o - 0.06) ** 2) * bg)[sl].mean()
                pk = (y > 0.5) & (y >= F.max_pool2d(y, 3, 1, 1))
                l = l + 5.0 * ((F.relu(0.6 - o) ** 2) * pk.float())[sl].sum() / max(int(pk[sl].sum()), 1)
                if hard_neg:
                    # hard-negative mining: the mean-squared background term above barely sees a handful of cross-type
                    # responses of 0.2-0.4 (ten cells in a 64 x 64 x 24 map); the strongest 0.05 % of the background cells of
                    # the batch are pushed below 0.05 directly -- what decides whether a false maximum crosses the 0.2 threshold
                    v = (F.relu(o - 0.05) * bg)[sl].flatten()
                    l = l + hard_neg * (v.topk(max(1, v.numel() // 2000)).values ** 2).mean()
                if t.get("nonneg"):
                    # confidence maps with negative lobes make integral refinement (centroid of a 5 x 5 patch = a division by
                    # the patch sum) ill-conditioned wherever the lobes cancel the peak: keep the maps (nearly) non-negative
                    l = l + float(t["nonneg"]) * (F.relu(-o) ** 2)[sl].mean()
            else:
                l = ((1.0 + 5.0 * y.abs()) * (o - y) ** 2)[sl].mean()
            parts.append(float(l.detach()))
            loss = loss + l
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        sched.step()
        if step % 50 == 0 or step == steps - 1:
            with torch.no_grad():
                o, y = outs[0], tg[0]
                pkv = (o * (y > 0.5)).amax(dim=(2, 3))
                has = (y > 0.5).flatten(2).any(-1)
                bgv = (o * (y < 0.01)).amax()
            print(f"[{task}] step {step:5d} loss {float(loss.detach()):.5f} {['%.5f' % p for p in parts]} peak@gt mean {float(pkv[has].mean()):.3f} "
                  f"min {float(pkv[has].min()):.3f} max bg {float(bgv):.3f} {time.time() - t0:.0f}s", flush=True)
        if (step + 1) % 500 == 0 or step == steps - 1:
            save(net, task, weights, out)
    print(f"[{task}] done {time.time() - t0:.0f}s", flush=True)


def save(net, task, weights, out):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    d = {}
    for k in weights:
        if k in net.frozen:
            continue
        d[k] = net.w(k).detach().cpu().numpy().astype(np.float32)
    if net.frozen:
        d["__frozen_checksum__"] = np.float64(sum(np.abs(weights[k].astype(np.float64)).sum() for k in sorted(net.frozen)))
    d["__task__"] = np.frombuffer(json.dumps({k: v for k, v in TASKS[task].items()}).encode(), dtype=np.uint8)
    np.savez(out, **d)
    print(f"[{task}] saved {out} {os.path.getsize(out)}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("tasks", nargs="+", choices=sorted(TASKS))
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--out-dir", default=None)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--hard-neg", type=float, default=0.0, help="weight of the hard-negative term (strongest 0.05 %% of background cells)")
    ap.add_argument("--bn-inference", action="store_true", help="BatchNormalization on the moving statistics from step 0 (resumed fits)")
    ap.add_argument("--crop", type=int, default=0, help="override the task's training crop size")
    a = ap.parse_args()
    for task in a.tasks:
        fit(task, a.steps, a.threads, lr=a.lr, resume=a.resume, device=a.device, out_dir=a.out_dir, batch=a.batch, hard_neg=a.hard_neg,
            bn_inference=a.bn_inference, crop=a.crop)
