"""Synthetic confidence-map / PAF generators (TEST INFRASTRUCTURE ONLY).

NumPy restatements of the reference's ground-truth map generators, used to build
inputs with known answers for the parity tests and the benchmark:
  sleap/nn/data/utils.py:41-84          make_grid_vectors, gaussian_pdf
  sleap/nn/data/confidence_maps.py:10-110  make_confmaps, make_multi_confmaps
  sleap/nn/data/edge_maps.py:16-211     distance_to_edge, make_edge_maps, make_pafs, make_multi_pafs
"""
import numpy as np

F32 = np.float32

# sleap/skeletons/flies13.json: 13 nodes / 12 edges
FLIES13_NODES = [
    "head", "thorax", "abdomen", "wingL", "wingR", "forelegL", "forelegR",
    "midlegL", "midlegR", "hindlegL", "hindlegR", "eyeL", "eyeR",
]
FLIES13_EDGES = [
    ("thorax", "head"), ("thorax", "abdomen"), ("thorax", "wingL"), ("thorax", "wingR"),
    ("thorax", "forelegL"), ("thorax", "forelegR"), ("thorax", "midlegL"), ("thorax", "midlegR"),
    ("thorax", "hindlegL"), ("thorax", "hindlegR"), ("head", "eyeL"), ("head", "eyeR"),
]
# template pose in body-length units, thorax at origin, head along +x
FLIES13_TEMPLATE = np.array(
    [
        [0.45, 0.0], [0.0, 0.0], [-0.55, 0.0], [-0.35, 0.30], [-0.35, -0.30],
        [0.35, 0.35], [0.35, -0.35], [0.05, 0.45], [0.05, -0.45],
        [-0.30, 0.50], [-0.30, -0.50], [0.55, 0.12], [0.55, -0.12],
    ],
    F32,
)


def make_grid_vectors(image_height, image_width, output_stride=1):
    """utils.py:41-70 -- `range(0, size, stride)` as float32."""
    xv = np.arange(0, image_width, output_stride).astype(F32)
    yv = np.arange(0, image_height, output_stride).astype(F32)
    return xv, yv


def gaussian_pdf(x, sigma):
    """utils.py:73-84 -- `exp(-x^2 / (2 sigma^2))`."""
    x = np.asarray(x, F32)
    return np.exp(-(x * x) / F32(2 * sigma * sigma)).astype(F32)


def make_confmaps(points, xv, yv, sigma):
    """confidence_maps.py:10-51 -- NaN points give all-zero channels."""
    points = np.asarray(points, F32).reshape(-1, 2)
    x = points[:, 0].reshape(1, 1, -1)
    y = points[:, 1].reshape(1, 1, -1)
    with np.errstate(invalid="ignore"):
        cm = np.exp(
            -((xv.reshape(1, -1, 1) - x) ** 2 + (yv.reshape(-1, 1, 1) - y) ** 2) / F32(2 * sigma ** 2)
        ).astype(F32)
    return np.where(np.isnan(cm), F32(0), cm)


def make_multi_confmaps(instances, xv, yv, sigma):
    """confidence_maps.py:57-110 -- max over instances having >= 1 node strictly inside."""
    instances = np.asarray(instances, F32).reshape(-1, np.asarray(instances).shape[-2], 2)
    cms = np.zeros((len(yv), len(xv), instances.shape[1]), F32)
    with np.errstate(invalid="ignore"):
        in_img = (instances > 0) & (instances < np.array([xv[-1], yv[-1]], F32).reshape(1, 1, 2))
    in_img = in_img.all(axis=-1).any(axis=1)
    for pts in instances[in_img]:
        cms = np.maximum(cms, make_confmaps(pts, xv, yv, sigma))
    return cms


def distance_to_edge(points, edge_source, edge_destination):
    """edge_maps.py:16-72 -- SQUARED distance from grid points to segments."""
    points = np.asarray(points, F32)  # (H, W, 2)
    src = np.asarray(edge_source, F32).reshape(-1, 2)
    dst = np.asarray(edge_destination, F32).reshape(-1, 2)
    direction = dst - src  # (E, 2)
    edge_len = np.maximum((direction ** 2).sum(axis=1), F32(1))
    rel = points[:, :, None, :] - src[None, None]  # (H, W, E, 2)
    proj = (rel * direction[None, None]).sum(axis=3) / edge_len[None, None]
    proj = np.clip(proj, 0, 1)
    d = ((proj[..., None] * direction[None, None] - rel) ** 2).sum(axis=3)
    return d.astype(F32)


def make_pafs(xv, yv, edge_source, edge_destination, sigma):
    """edge_maps.py:119-162 -- `gaussian_pdf(squared distance) * unit vector` (reference quirk:
    the already-squared distance is squared again inside gaussian_pdf)."""
    src = np.asarray(edge_source, F32).reshape(-1, 2)
    dst = np.asarray(edge_destination, F32).reshape(-1, 2)
    with np.errstate(invalid="ignore", divide="ignore"):
        unit = dst - src
        unit = unit / np.sqrt((unit ** 2).sum(axis=-1, keepdims=True))
    grid = np.stack(np.meshgrid(xv, yv), axis=-1)  # (H, W, 2)
    em = gaussian_pdf(distance_to_edge(grid, src, dst), sigma)  # (H, W, E)
    return (em[..., None] * unit[None, None]).astype(F32)


def make_multi_pafs(xv, yv, edge_sources, edge_destinations, sigma):
    """edge_maps.py:165-211 -- SUM over instances, NaNs dropped; returns (H, W, E, 2)."""
    edge_sources = np.asarray(edge_sources, F32)
    edge_destinations = np.asarray(edge_destinations, F32)
    pafs = np.zeros((len(yv), len(xv), edge_sources.shape[1], 2), F32)
    for i in range(edge_sources.shape[0]):
        paf = make_pafs(xv, yv, edge_sources[i], edge_destinations[i], sigma)
        pafs += np.where(np.isnan(paf), F32(0), paf)
    return pafs


def random_fly_instances(rng, n_animals, height, width, body=(80.0, 120.0), margin=128.0,
                         min_sep=64.0, jitter=2.0, template=FLIES13_TEMPLATE):
    """Seeded random similarity transforms of the template pose (SURVEY.md §8d generator spec)."""
    centres = []
    out = []
    tries = 0
    while len(out) < n_animals and tries < 10000:
        tries += 1
        c = rng.uniform([margin, margin], [width - margin, height - margin])
        if any(np.hypot(*(c - o)) < min_sep for o in centres):
            continue
        th = rng.uniform(0, 2 * np.pi)
        s = rng.uniform(*body)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        pts = (template * s) @ R.T + c + rng.normal(0, jitter, template.shape)
        centres.append(c)
        out.append(pts.astype(F32))
    return np.stack(out).astype(F32) if out else np.zeros((0, template.shape[0], 2), F32)


def synth_bottomup_maps(instances, height, width, node_names=FLIES13_NODES, edges=FLIES13_EDGES,
                        cm_stride=4, paf_stride=8, cm_sigma=2.5, paf_sigma=75.0, noise=0.01, rng=None):
    """Analytic cms (H/cm_stride, W/cm_stride, N) and PAFs (H/paf_stride, W/paf_stride, 2E).

    Generators work in image-pixel coordinates because the grid vectors are 0, stride, 2*stride...
    PAF channels are reshaped (H, W, E, 2) -> (H, W, 2E), x-component at 2k (heads.py:281-283).
    """
    edge_inds = [(node_names.index(s), node_names.index(d)) for s, d in edges]
    xv, yv = make_grid_vectors(height, width, cm_stride)
    cms = make_multi_confmaps(instances, xv, yv, cm_sigma)
    xv8, yv8 = make_grid_vectors(height, width, paf_stride)
    src = instances[:, [e[0] for e in edge_inds], :]
    dst = instances[:, [e[1] for e in edge_inds], :]
    pafs = make_multi_pafs(xv8, yv8, src, dst, paf_sigma)
    pafs = pafs.reshape(pafs.shape[0], pafs.shape[1], -1)
    if noise and rng is not None:
        cms = cms + rng.normal(0, noise, cms.shape).astype(F32)
        pafs = pafs + rng.normal(0, noise, pafs.shape).astype(F32)
    return cms.astype(F32), pafs.astype(F32), edge_inds


def render_fly_frames(instances_per_frame, height, width, rng, edges=FLIES13_EDGES,
                      node_names=FLIES13_NODES):
    """uint8 frames (T, H, W, 1): smooth noise background with dark blobs on nodes and limbs."""
    T = len(instances_per_frame)
    frames = np.empty((T, height, width, 1), np.uint8)
    yy, xx = np.mgrid[0:height, 0:width].astype(F32)
    edge_inds = [(node_names.index(s), node_names.index(d)) for s, d in edges]
    for t, inst in enumerate(instances_per_frame):
        small = rng.normal(0, 1, (height // 16 + 1, width // 16 + 1)).astype(F32)
        bg = np.kron(small, np.ones((16, 16), F32))[:height, :width]
        img = 170 + 12 * bg + rng.normal(0, 4, (height, width)).astype(F32)
        for a in inst:
            for (s, d) in edge_inds:
                p0, p1 = a[s], a[d]
                n = max(int(np.hypot(*(p1 - p0)) / 2), 2)
                for u in np.linspace(0, 1, n):
                    p = p0 * (1 - u) + p1 * u
                    x0, x1 = int(max(p[0] - 6, 0)), int(min(p[0] + 7, width))
                    y0, y1 = int(max(p[1] - 6, 0)), int(min(p[1] + 7, height))
                    if x1 <= x0 or y1 <= y0:
                        continue
                    d2 = (xx[y0:y1, x0:x1] - p[0]) ** 2 + (yy[y0:y1, x0:x1] - p[1]) ** 2
                    img[y0:y1, x0:x1] -= 60 * np.exp(-d2 / (2 * 2.0 ** 2)) * 0.25
            for k, p in enumerate(a):
                x0, x1 = int(max(p[0] - 9, 0)), int(min(p[0] + 10, width))
                y0, y1 = int(max(p[1] - 9, 0)), int(min(p[1] + 10, height))
                if x1 <= x0 or y1 <= y0:
                    continue
                d2 = (xx[y0:y1, x0:x1] - p[0]) ** 2 + (yy[y0:y1, x0:x1] - p[1]) ** 2
                img[y0:y1, x0:x1] -= (70 + 5 * k) * np.exp(-d2 / (2 * 3.0 ** 2))
        frames[t, :, :, 0] = np.clip(img, 0, 255).astype(np.uint8)
    return frames
