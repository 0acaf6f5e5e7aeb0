"""Shared comparison of a device bottom-up result with the oracle's when the maps hold borderline local maxima."""
import numpy as np


def compare_with_threshold_decisions(oracle_peaks, device_peaks, ref, o, n_nodes, map_eps, tol_px, threshold=0.2, ill=None):
    """oracle_peaks = (pts (n, 2) image px, vals, sample_inds, channel_inds) of find_local_peaks; device_peaks = (peak_xy [B, P, 2],
    peak_val, peak_chan, peak_count) of the device layer; ref = the oracle's PAFScorer.predict result; o = the device layer's
    outputs (numpy). Asserts:

      * every peak only ONE path detects has a confidence within `map_eps` of the threshold (a threshold decision on a map
        value that differs by the storage precision) -- nothing else may differ;
      * frames whose peak sets agree give the same instances: count, node assignment, every coordinate within `tol_px`.

    `ill`: boolean mask over the oracle's peaks whose REFINEMENT is ill-conditioned (integral regression divides by the sum of a
    5 x 5 patch; where the patch holds negative values next to positive ones the sum is ~0 and the "refined" position lands
    tens to thousands of pixels away -- in the reference too). Such a peak is only required to EXIST on the device (same
    channel, confidence within `map_eps`); its coordinates are not compared, and its frame is not compared at instance level.

    -> (frames whose peak sets differ or hold ill-conditioned peaks, number of common peaks, their largest distance, number of
    one-sided peaks)."""
    pts, vals, si, ci = oracle_peaks
    g_xy, g_val, g_ch, g_n = device_peaks
    n = len(g_n)
    differing, n_common, worst, n_only = [], 0, 0.0, 0
    for b in range(n):
        wp, wv, wc = pts[si == b], vals[si == b], ci[si == b]
        gp, gv, gc = g_xy[b, : g_n[b]], g_val[b, : g_n[b]], g_ch[b, : g_n[b]]
        used = np.zeros(len(gp), bool)
        same = True
        wi = ill[si == b] if ill is not None else np.zeros(len(wp), bool)
        for p, v, c in zip(wp[wi], wv[wi], wc[wi]):  # ill-conditioned refinements first: matched by channel and confidence
            cand = np.where((gc == c) & ~used & (np.abs(gv - v) <= map_eps))[0]
            assert len(cand), f"frame {b}: the device has no peak of channel {c} with confidence {v} (ill-conditioned refinement)"
            used[cand[int(np.abs(gv[cand] - v).argmin())]] = True
            same = False
        wp, wv, wc = wp[~wi], wv[~wi], wc[~wi]
        for p, v, c in zip(wp, wv, wc):
            cand = np.where((gc == c) & ~used)[0]
            d = np.linalg.norm(gp[cand] - p, axis=-1) if len(cand) else np.zeros(0)
            if len(cand) and d.min() <= 2.0:  # the same local maximum (grid cells are >= 2 px apart)
                j = cand[int(d.argmin())]
                used[j] = True
                n_common += 1
                worst = max(worst, float(d.min()))
            else:
                same = False
                n_only += 1
                assert abs(float(v) - threshold) <= map_eps, f"frame {b}: oracle-only peak with value {v} (channel {c})"
        for j in np.where(~used)[0]:
            same = False
            n_only += 1
            assert abs(float(gv[j]) - threshold) <= map_eps, f"frame {b}: device-only peak with value {gv[j]} (channel {gc[j]})"
        if not same:
            differing.append(b)
            continue
        want = np.asarray(ref[0][b]).reshape(-1, n_nodes, 2)
        assert int(o["n_valid"][b]) == len(want), f"frame {b}: same peaks, different instance count"
        got = o["instance_peaks"][b, : len(want)]
        assert np.array_equal(np.isnan(got), np.isnan(want)), f"frame {b}: same peaks, different node assignment"
        if np.isfinite(got).any():
            assert float(np.nanmax(np.linalg.norm(got - want, axis=-1))) <= tol_px
    return differing, n_common, worst, n_only


def integral_refinement_is_ill_conditioned(cms, refined_grid, rough_grid, sample_inds, channel_inds, min_ratio=0.6, max_offset=1.0):
    """Which peaks of find_local_peaks(cms, refinement="integral") (grid units, before the stride multiplication) have an
    ill-conditioned refinement: integral regression (peak_finding.py:78-132) is the centroid of a 5 x 5 patch, i.e. a division by
    the patch SUM -- where the map has negative lobes around a maximum the sum is small against the sum of magnitudes and the
    quotient amplifies any difference in the map values (the reference behaves the same way). Flagged: patch sum below
    `min_ratio` of the sum of magnitudes, or a centroid further than `max_offset` cells from the patch centre."""
    cp = np.pad(np.asarray(cms), ((0, 0), (2, 2), (2, 2), (0, 0)))
    ratio = np.empty(len(rough_grid))
    for k, (p, b, c) in enumerate(zip(rough_grid, sample_inds, channel_inds)):
        x, y = int(p[0]), int(p[1])
        patch = cp[b, y:y + 5, x:x + 5, c]
        ratio[k] = patch.sum() / max(float(np.abs(patch).sum()), 1e-30)
    off = np.abs(np.asarray(refined_grid) - np.asarray(rough_grid)).max(axis=1)
    return (ratio < min_ratio) | (off > max_offset)
