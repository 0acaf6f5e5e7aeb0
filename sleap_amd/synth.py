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


def random_instances(rng, n_animals, height, width, body=(80.0, 120.0), margin=128.0, min_sep=64.0, jitter=2.0,
                     template=None):
    template = _TEMPLATE if template is None else template
    centres, out, tries = [], [], 0
    while len(out) < n_animals and tries < 10000:
        tries += 1
        c = rng.uniform([margin, margin], [width - margin, height - margin])
        if any(np.hypot(*(c - o)) < min_sep for o in centres):
            continue
        th, s = rng.uniform(0, 2 * np.pi), rng.uniform(*body)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        out.append(((template * s) @ R.T + c + rng.normal(0, jitter, template.shape)).astype(np.float32))
        centres.append(c)
    return np.stack(out) if out else np.zeros((0, len(template), 2), np.float32)


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
# per node: (shape, amplitude subtracted from the ~170 background, radius in px). Body parts are large dark discs, wings large
# faint ones, legs small discs (left 60 / right 120) except the hind legs, which are rings, and the eyes are the only BRIGHT
# marks (they saturate at 255, which nothing in the background reaches): left / right and fore / mid / hind are told apart by
# amplitude steps of >= 35 grey levels (pixel noise: 4) or by shape, never by context alone.
_NODE_CODE = [("dot", 150.0, 4.5), ("dot", 165.0, 6.5), ("dot", 135.0, 5.5), ("dot", 45.0, 6.0), ("dot", 85.0, 6.0),
              ("dot", 60.0, 2.3), ("dot", 120.0, 2.3), ("dot", 60.0, 3.6), ("dot", 120.0, 3.6), ("ring", 70.0, 3.5),
              ("ring", 130.0, 3.5), ("dot", -85.0, 2.0), ("dot", -85.0, 3.5)]


class Skeleton:
    """A rendered animal type: node names, edges (by name), template pose (unit body length) and the per-node mark code."""

    def __init__(self, nodes, edges, template, node_code):
        self.nodes, self.edges = list(nodes), [tuple(e) for e in edges]
        self.template = np.asarray(template, np.float32)
        self.node_code = list(node_code)
        self.edge_idx = [(self.nodes.index(a), self.nodes.index(b)) for a, b in self.edges]
        assert len(self.nodes) == len(self.template) == len(self.node_code)


FLIES13 = Skeleton(FLIES13_NODES, FLIES13_EDGES, _TEMPLATE, _NODE_CODE)
# configs[0]: the 5 body nodes of the fly (head, thorax, abdomen, wings) -- same marks, same template
FLIES5 = Skeleton(FLIES13_NODES[:5], FLIES13_EDGES[:4], _TEMPLATE[:5], _NODE_CODE[:5])

# configs[4]: a 24-node / 23-edge "mouse" (BASELINE.json names the node count only; a tree): a 12-node spine from
# nose to tail tip, ears and eyes on the head, four two-segment limbs. Marks: every node has its own (shape, amplitude,
# radius) triple; left / right partners differ by >= 40 grey levels or by shape.
MOUSE24_NODES = ["nose", "head", "neck", "spine1", "spine2", "spine3", "spine4", "tailbase", "tail1", "tail2", "tail3", "tailtip",
                 "earL", "earR", "eyeL", "eyeR", "shoulderL", "pawFL", "shoulderR", "pawFR", "hipL", "pawHL", "hipR", "pawHR"]
MOUSE24_EDGES = [("head", "nose"), ("head", "neck"), ("neck", "spine1"), ("spine1", "spine2"), ("spine2", "spine3"),
                 ("spine3", "spine4"), ("spine4", "tailbase"), ("tailbase", "tail1"), ("tail1", "tail2"), ("tail2", "tail3"),
                 ("tail3", "tailtip"), ("head", "earL"), ("head", "earR"), ("head", "eyeL"), ("head", "eyeR"),
                 ("spine1", "shoulderL"), ("shoulderL", "pawFL"), ("spine1", "shoulderR"), ("shoulderR", "pawFR"),
                 ("spine4", "hipL"), ("hipL", "pawHL"), ("spine4", "hipR"), ("hipR", "pawHR")]
_MOUSE_TEMPLATE = np.array([[0.62, 0.0], [0.46, 0.0], [0.30, 0.0], [0.14, 0.0], [-0.02, 0.0], [-0.18, 0.0], [-0.34, 0.0],
                            [-0.50, 0.0], [-0.66, 0.03], [-0.82, 0.08], [-0.97, 0.15], [-1.10, 0.24],
                            [0.40, 0.17], [0.40, -0.17], [0.55, 0.08], [0.55, -0.08],
                            [0.16, 0.20], [0.24, 0.38], [0.16, -0.20], [0.24, -0.38],
                            [-0.34, 0.22], [-0.28, 0.42], [-0.34, -0.22], [-0.28, -0.42]], np.float32)
_MOUSE_CODE = [("dot", 120.0, 2.5), ("dot", 165.0, 6.0), ("dot", 100.0, 4.0), ("dot", 150.0, 5.0), ("ring", 120.0, 4.5),
               ("dot", 135.0, 6.5), ("ring", 150.0, 3.0), ("dot", 165.0, 4.0), ("dot", 90.0, 3.0), ("ring", 90.0, 3.0),
               ("dot", 60.0, 2.5), ("dot", -85.0, 2.5),
               ("dot", 50.0, 5.0), ("dot", 110.0, 5.0), ("dot", -85.0, 1.8), ("ring", -85.0, 3.0),
               ("dot", 70.0, 3.5), ("dot", 60.0, 2.3), ("dot", 130.0, 3.5), ("dot", 120.0, 2.3),
               ("ring", 70.0, 4.0), ("dot", 45.0, 3.2), ("ring", 130.0, 4.0), ("dot", 105.0, 3.2)]
MOUSE24 = Skeleton(MOUSE24_NODES, MOUSE24_EDGES, _MOUSE_TEMPLATE, _MOUSE_CODE)


def render_animals(n_frames, height, width, n_animals=4, seed=0, min_sep=170.0, margin=128.0, skeleton=FLIES13,
                   body=(80.0, 120.0), noise=4.0, contrast=1.0):
    """-> (frames uint8 (T, H, W, 1), list of (A, N, 2) float32 instance arrays in (x, y) image pixels).

    `noise` = sigma of the per-pixel noise (grey levels), `contrast` scales every mark's amplitude: the parity tests' "hard"
    variants lower the contrast / raise the noise so that detections approach the peak threshold."""
    rng = np.random.default_rng(seed)
    frames = np.empty((n_frames, height, width, 1), np.uint8)
    insts = []
    for t in range(n_frames):
        inst = random_instances(rng, n_animals, height, width, body=body, margin=min(margin, min(height, width) / 4),
                                min_sep=min_sep, template=skeleton.template)
        insts.append(inst)
        small = rng.normal(0, 1, (height // 16 + 1, width // 16 + 1)).astype(np.float32)
        img = 170 + 12 * np.kron(small, np.ones((16, 16), np.float32))[:height, :width]
        img += rng.normal(0, noise, (height, width)).astype(np.float32)
        for a in inst:
            for (s, d) in skeleton.edge_idx:  # limbs first, node blobs on top
                p0, p1 = a[s].astype(np.float64), a[d].astype(np.float64)
                x0, x1 = int(max(min(p0[0], p1[0]) - 6, 0)), int(min(max(p0[0], p1[0]) + 7, width))
                y0, y1 = int(max(min(p0[1], p1[1]) - 6, 0)), int(min(max(p0[1], p1[1]) + 7, height))
                if x1 <= x0 or y1 <= y0:
                    continue
                yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
                v = p1 - p0
                tt = np.clip(((xx - p0[0]) * v[0] + (yy - p0[1]) * v[1]) / max(float(v @ v), 1.0), 0.0, 1.0)
                d2 = (xx - (p0[0] + tt * v[0])) ** 2 + (yy - (p0[1] + tt * v[1])) ** 2
                img[y0:y1, x0:x1] -= contrast * (55.0 - 35.0 * tt) * np.exp(-d2 / (2 * 1.5 ** 2))
            for k, p in enumerate(a):
                shape, amp, rad = skeleton.node_code[k]
                r = int(3 * rad) + 2
                cx, cy = int(round(float(p[0]))), int(round(float(p[1])))
                x0, x1, y0, y1 = max(cx - r, 0), min(cx + r + 1, width), max(cy - r, 0), min(cy + r + 1, height)
                if x1 <= x0 or y1 <= y0:
                    continue
                yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
                d2 = (xx - p[0]) ** 2 + (yy - p[1]) ** 2
                if shape == "ring":
                    img[y0:y1, x0:x1] -= contrast * amp * np.exp(-((np.sqrt(d2) - rad) ** 2) / (2 * 1.2 ** 2))
                else:
                    img[y0:y1, x0:x1] -= contrast * amp * np.exp(-d2 / (2 * rad ** 2))
        frames[t, :, :, 0] = np.clip(img, 0, 255).astype(np.uint8)
    return frames, insts


def render_flies(n_frames, height, width, n_animals=4, seed=0, min_sep=170.0, margin=128.0, return_instances=True):
    """The 13-node fly video (the benchmark model was fitted to it): `render_animals` with the FLIES13 skeleton."""
    return render_animals(n_frames, height, width, n_animals, seed, min_sep, margin, FLIES13)
