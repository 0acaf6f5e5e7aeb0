"""Seeded synthetic 13-node fly video for benchmarks and smoke tests (no datasets are reachable offline).

Skeleton = sleap/skeletons/flies13.json (13 nodes / 12 edges); generator spec from SURVEY.md §8(d):
random similarity transforms of a template pose, centres kept `margin` px inside the frame and at
least `min_sep` px apart.
"""
import numpy as np

FLIES13_NODES = ["head", "thorax", "abdomen", "wingL", "wingR", "forelegL", "forelegR", "midlegL", "midlegR",
                 "hindlegL", "hindlegR", "eyeL", "eyeR"]
FLIES13_EDGES = [("thorax", "head"), ("thorax", "abdomen"), ("thorax", "wingL"), ("thorax", "wingR"),
                 ("thorax", "forelegL"), ("thorax", "forelegR"), ("thorax", "midlegL"), ("thorax", "midlegR"),
                 ("thorax", "hindlegL"), ("thorax", "hindlegR"), ("head", "eyeL"), ("head", "eyeR")]
_TEMPLATE = np.array([[0.45, 0.0], [0.0, 0.0], [-0.55, 0.0], [-0.35, 0.30], [-0.35, -0.30], [0.35, 0.35], [0.35, -0.35],
                      [0.05, 0.45], [0.05, -0.45], [-0.30, 0.50], [-0.30, -0.50], [0.55, 0.12], [0.55, -0.12]], np.float32)


def random_instances(rng, n_animals, height, width, body=(80.0, 120.0), margin=128.0, min_sep=64.0, jitter=2.0):
    centres, out, tries = [], [], 0
    while len(out) < n_animals and tries < 10000:
        tries += 1
        c = rng.uniform([margin, margin], [width - margin, height - margin])
        if any(np.hypot(*(c - o)) < min_sep for o in centres):
            continue
        th, s = rng.uniform(0, 2 * np.pi), rng.uniform(*body)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        out.append(((_TEMPLATE * s) @ R.T + c + rng.normal(0, jitter, _TEMPLATE.shape)).astype(np.float32))
        centres.append(c)
    return np.stack(out) if out else np.zeros((0, 13, 2), np.float32)


def render_frames(n_frames, height, width, n_animals=4, seed=0):
    """-> (frames uint8 (T, H, W, 1), instances list of (A, 13, 2)). Dark blobs on a noisy light background."""
    rng = np.random.default_rng(seed)
    frames = np.empty((n_frames, height, width, 1), np.uint8)
    insts = []
    r = 10
    yy, xx = np.mgrid[-r : r + 1, -r : r + 1].astype(np.float32)
    for t in range(n_frames):
        inst = random_instances(rng, n_animals, height, width, margin=min(128.0, min(height, width) / 4))
        insts.append(inst)
        small = rng.normal(0, 1, (height // 16 + 1, width // 16 + 1)).astype(np.float32)
        img = 170 + 12 * np.kron(small, np.ones((16, 16), np.float32))[:height, :width]
        img += rng.normal(0, 4, (height, width)).astype(np.float32)
        for a in inst:
            for k, p in enumerate(a):
                cx, cy = int(round(float(p[0]))), int(round(float(p[1])))
                x0, x1, y0, y1 = max(cx - r, 0), min(cx + r + 1, width), max(cy - r, 0), min(cy + r + 1, height)
                if x1 <= x0 or y1 <= y0:
                    continue
                blob = np.exp(-((xx + cx - p[0]) ** 2 + (yy + cy - p[1]) ** 2) / (2 * 3.0 ** 2))
                img[y0:y1, x0:x1] -= (70 + 5 * k) * blob[y0 - cy + r : y1 - cy + r, x0 - cx + r : x1 - cx + r]
        frames[t, :, :, 0] = np.clip(img, 0, 255).astype(np.uint8)
    return frames, insts


# ---------------------------------------------------------------------------------------------------------------------------
# v2 renderer: frames a network can actually be TRAINED on (tools/train_benchmark_model.py) so that the benchmark and the
# end-to-end parity tests run on confident, well separated detections (4 instances x 13 nodes per frame) instead of the
# noise-like maps of a random-init network.
#   * every node type has its own blob (amplitude / radius code), so node identity is decodable locally;
#   * every skeleton edge is drawn as a line that fades from its source to its destination node, so the direction of the
#     part-affinity field is decodable locally;
#   * animals keep `min_sep` px between centres (default 170 > the largest body extent), i.e. they never overlap.
# Same skeleton, template pose, similarity transforms, jitter and background statistics as `render_frames`.
# ---------------------------------------------------------------------------------------------------------------------------
_EDGE_IDX = [(FLIES13_NODES.index(a), FLIES13_NODES.index(b)) for a, b in FLIES13_EDGES]
# per node: (shape, amplitude subtracted from the ~170 background, radius in px). Body parts are large dark discs, wings large
# faint ones, legs small discs (left 60 / right 120) except the hind legs, which are rings, and the eyes are the only BRIGHT
# marks (they saturate at 255, which nothing in the background reaches): left / right and fore / mid / hind are told apart by
# amplitude steps of >= 35 grey levels (pixel noise: 4) or by shape, never by context alone.
_NODE_CODE = [("dot", 150.0, 4.5), ("dot", 165.0, 6.5), ("dot", 135.0, 5.5), ("dot", 45.0, 6.0), ("dot", 85.0, 6.0),
              ("dot", 60.0, 2.3), ("dot", 120.0, 2.3), ("dot", 60.0, 3.6), ("dot", 120.0, 3.6), ("ring", 70.0, 3.5),
              ("ring", 130.0, 3.5), ("dot", -85.0, 2.0), ("dot", -85.0, 3.5)]


def render_flies(n_frames, height, width, n_animals=4, seed=0, min_sep=170.0, margin=128.0, return_instances=True):
    """-> (frames uint8 (T, H, W, 1), list of (A, 13, 2) float32 instance arrays in (x, y) image pixels)."""
    rng = np.random.default_rng(seed)
    frames = np.empty((n_frames, height, width, 1), np.uint8)
    insts = []
    for t in range(n_frames):
        inst = random_instances(rng, n_animals, height, width, margin=min(margin, min(height, width) / 4), min_sep=min_sep)
        insts.append(inst)
        small = rng.normal(0, 1, (height // 16 + 1, width // 16 + 1)).astype(np.float32)
        img = 170 + 12 * np.kron(small, np.ones((16, 16), np.float32))[:height, :width]
        img += rng.normal(0, 4, (height, width)).astype(np.float32)
        for a in inst:
            for (s, d) in _EDGE_IDX:  # limbs first, node blobs on top
                p0, p1 = a[s].astype(np.float64), a[d].astype(np.float64)
                x0, x1 = int(max(min(p0[0], p1[0]) - 6, 0)), int(min(max(p0[0], p1[0]) + 7, width))
                y0, y1 = int(max(min(p0[1], p1[1]) - 6, 0)), int(min(max(p0[1], p1[1]) + 7, height))
                if x1 <= x0 or y1 <= y0:
                    continue
                yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
                v = p1 - p0
                tt = np.clip(((xx - p0[0]) * v[0] + (yy - p0[1]) * v[1]) / max(float(v @ v), 1.0), 0.0, 1.0)
                d2 = (xx - (p0[0] + tt * v[0])) ** 2 + (yy - (p0[1] + tt * v[1])) ** 2
                img[y0:y1, x0:x1] -= (55.0 - 35.0 * tt) * np.exp(-d2 / (2 * 1.5 ** 2))
            for k, p in enumerate(a):
                shape, amp, rad = _NODE_CODE[k]
                r = int(3 * rad) + 2
                cx, cy = int(round(float(p[0]))), int(round(float(p[1])))
                x0, x1, y0, y1 = max(cx - r, 0), min(cx + r + 1, width), max(cy - r, 0), min(cy + r + 1, height)
                if x1 <= x0 or y1 <= y0:
                    continue
                yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
                d2 = (xx - p[0]) ** 2 + (yy - p[1]) ** 2
                if shape == "ring":
                    img[y0:y1, x0:x1] -= amp * np.exp(-((np.sqrt(d2) - rad) ** 2) / (2 * 1.2 ** 2))
                else:
                    img[y0:y1, x0:x1] -= amp * np.exp(-d2 / (2 * rad ** 2))
        frames[t, :, :, 0] = np.clip(img, 0, 255).astype(np.uint8)
    return frames, insts
