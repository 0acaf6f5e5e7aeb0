"""What the fp32 ORACLE sees on fresh frames with the fitted config models (tests/config_models.py): peak values vs the 0.2
threshold, background maxima, distance to the rendered truth. Diagnostic, CPU only:  python tests/diagnostics/config_models_eval.py [task ...]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")  # config_models.py: test infrastructure
from oracle import inference as oinf  # noqa: E402
from oracle import paf_grouping as opg  # noqa: E402
from oracle import peak_finding as opf  # noqa: E402
from oracle.keras_graph import KerasGraph, preprocess  # noqa: E402
import config_models as C  # noqa: E402


def single(task, n=8, seed=300):
    t = C.TASKS[task]
    frames, insts = C.render(task, n, seed)
    x = preprocess(frames, input_scale=t["input_scale"], pad_stride=t["unet"][2])
    mc, w = C.load_task_weights(task, x.shape[1], x.shape[2])
    cms = KerasGraph(mc, w)(x)[0]
    pk, vals = oinf.single_instance_peaks(cms, None, 0.2, "integral", 5, t["heads"][0][2], t["input_scale"])
    gt = np.stack([a[0] for a in insts])
    d = np.linalg.norm(pk[:, 0] - gt, axis=-1)
    bg = cms.copy()
    print(f"{task}: peak vals min {np.nanmin(vals):.3f} mean {np.nanmean(vals):.3f}; NaN peaks {int(np.isnan(pk[..., 0]).sum())}; "
          f"dist to truth mean {np.nanmean(d):.2f} max {np.nanmax(d):.2f} px; map max {bg.max():.3f}")


def topdown(n=4, seed=300):
    tc, ti = C.TASKS["c2_centroid"], C.TASKS["c2_centered"]
    frames, insts = C.render("c2_centroid", n, seed)
    x = preprocess(frames, input_scale=tc["input_scale"], pad_stride=16)
    mc, w = C.load_task_weights("c2_centroid", x.shape[1], x.shape[2])
    cms = KerasGraph(mc, w)(x)[0]
    cc = oinf.centroid_crop(frames, cms, None, 0.2, "integral", 5, 2, 0.5, ti["crop"])
    print(f"c2_centroid: {len(cc['centroids'])} centroids in {n} frames (truth {sum(len(a) for a in insts)}), vals "
          f"{cc['centroid_vals'].min():.3f}..{cc['centroid_vals'].max():.3f}; background max "
          f"{np.sort(cms.reshape(n, -1), axis=1)[:, -200].max():.3f}")
    gt = np.concatenate([a[:, C.ANCHOR] for a in insts])
    dd = np.linalg.norm(cc["centroids"][:, None] - gt[None], axis=-1).min(axis=1)
    print(f"             distance centroid -> nearest thorax: mean {dd.mean():.2f} max {dd.max():.2f} px")
    mc2, w2 = C.load_task_weights("c2_centered", ti["crop"], ti["crop"])
    crops = preprocess(cc["crops"])
    cm2 = KerasGraph(mc2, w2)(crops)[0]
    pk, vals = oinf.find_instance_peaks(cm2, None, cc["crop_offsets"], 0.2, "integral", 5, 4, 1.0)
    allgt = np.concatenate(insts)
    j = np.linalg.norm(cc["centroids"][:, None] - allgt[None, :, C.ANCHOR], axis=-1).argmin(axis=1)
    d = np.linalg.norm(pk - allgt[j], axis=-1)
    print(f"c2_centered: peak vals min {np.nanmin(vals):.3f} mean {np.nanmean(vals):.3f}; NaN {int(np.isnan(pk[..., 0]).sum())} of {pk[..., 0].size}; "
          f"dist to truth mean {np.nanmean(d):.2f} max {np.nanmax(d):.2f} px")


def bottomup(task="c4_resnet", n=2, seed=300):
    t = C.TASKS[task]
    sk = C.skeleton(task)
    frames, insts = C.render(task, n, seed)
    mc, w = C.load_task_weights(task, frames.shape[1], frames.shape[2])
    cms, pafs = KerasGraph(mc, w)(preprocess(frames))[:2]
    pts, vals, si, ci = opf.find_local_peaks(cms, 0.2, "integral", 5)
    pts = pts * np.float32(4)
    sc = opg.PAFScorer(sk.nodes, sk.edges, 8, oob="zero")
    ref = sc.predict(pafs, [pts[si == b] for b in range(n)], [vals[si == b] for b in range(n)], [ci[si == b] for b in range(n)])
    for b in range(n):
        inst = np.asarray(ref[0][b]).reshape(-1, len(sk.nodes), 2)
        full = int((~np.isnan(inst[..., 0])).all(axis=1).sum())
        print(f"{task} frame {b}: {int((si == b).sum())} peaks (truth {insts[b].shape[0] * insts[b].shape[1]}), {len(inst)} instances, "
              f"{full} complete; peak vals min {vals[si == b].min():.3f}; scores {np.round(np.asarray(ref[2][b]), 2).tolist()}")
    v = np.sort(vals)
    print(f"{task}: weakest peaks {np.round(v[:6], 3).tolist()}; cms max {cms.max():.3f}")


if __name__ == "__main__":
    todo = sys.argv[1:] or ["c0_single5", "c1_single13", "topdown", "c4_resnet"]
    for t in todo:
        if t == "topdown":
            topdown()
        elif t == "c4_resnet":
            bottomup()
        else:
            single(t)
