"""Shared comparison of a device bottom-up result with the oracle's when the maps hold borderline local maxima."""
import numpy as np


def compare_with_threshold_decisions(oracle_peaks, device_peaks, ref, o, n_nodes, map_eps, tol_px, threshold=0.2, ill=None,
                                     cms=None, stride=4, strict_instances=True, stats=None, rough_grid=None):
    """oracle_peaks = (pts (n, 2) image px, vals, sample_inds, channel_inds) of find_local_peaks; device_peaks = (peak_xy [B, P, 2],
    peak_val, peak_chan, peak_count) of the device layer; ref = the oracle's PAFScorer.predict result; o = the device layer's
    outputs (numpy). The two paths compute the same maps up to the storage precision (`map_eps`); every DECISION taken on
    nearly equal numbers can therefore legitimately differ, and nothing else may. Asserted:

      * every peak both paths detect (same channel, within 2 px) agrees -- the largest distance is returned for the caller's
        0.5 px assertion;
      * a peak only ONE path detects has a confidence within `map_eps` of the threshold (a threshold decision);
      * NEAR TIE (needs the oracle's maps `cms`): the device's maximum sits in a NEIGHBOURING grid cell of the oracle's and the
        oracle's own map values at the two cells differ by <= `map_eps` (which of two nearly equal cells is "the" local maximum
        is decided by the last bits; the refined positions then differ by up to a cell). Counted, not compared;
      * RIDGE (needs `cms` and the oracle's grid maxima `rough_grid`): an oracle maximum that exceeds one of its eight neighbours
        by <= 2 `map_eps` need not be a strict local maximum of the device's map at all (and vice versa) -- excused, counted
        with the near ties;
      * `ill`: boolean mask over the oracle's peaks whose REFINEMENT is ill-conditioned (integral regression divides by the
        sum of a 5 x 5 patch; where negative lobes cancel the peak the "refined" position lands anywhere -- in the reference
        too). Such a peak is only required to EXIST on the device (same channel, confidence within `map_eps`);
      * frames without any such difference give the same instances: count, node assignment, every coordinate within `tol_px`.

    -> (frames not compared at instance level, number of common peaks, their largest distance, number of one-sided peaks,
    number of near ties)."""
    pts, vals, si, ci = oracle_peaks
    g_xy, g_val, g_ch, g_n = device_peaks
    n = len(g_n)
    differing, n_common, worst, n_only, n_tie = [], 0, 0.0, 0, 0
    for b in range(n):
        m = si == b
        wp, wv, wc = pts[m], vals[m], ci[m]
        wi = ill[m] if ill is not None else np.zeros(len(wp), bool)
        gp, gv, gc = g_xy[b, : g_n[b]], g_val[b, : g_n[b]], g_ch[b, : g_n[b]]
        used = np.zeros(len(gp), bool)
        same = True
        pending = []
        for k in np.where(~wi)[0]:  # well-conditioned peaks: by position
            p, c = wp[k], wc[k]
            cand = np.where((gc == c) & ~used)[0]
            d = np.linalg.norm(gp[cand] - p, axis=-1) if len(cand) else np.zeros(0)
            if len(cand) and d.min() <= 2.0:  # the same local maximum (grid cells are >= 2 px apart)
                used[cand[int(d.argmin())]] = True
                n_common += 1
                worst = max(worst, float(d.min()))
            else:
                pending.append(k)
        for k in pending:
            p, v, c = wp[k], wv[k], wc[k]
            same = False
            cand = np.where((gc == c) & ~used)[0]
            d = np.linalg.norm(gp[cand] - p, axis=-1) if len(cand) else np.zeros(0)
            if cms is not None and len(cand) and d.min() <= 1.6 * stride:
                j = cand[int(d.argmin())]
                x0, y0 = int(round(float(p[0]) / stride)), int(round(float(p[1]) / stride))
                x1, y1 = int(round(float(gp[j][0]) / stride)), int(round(float(gp[j][1]) / stride))
                hh, ww = cms.shape[1:3]
                inside = all(0 <= q < lim for q, lim in ((x0, ww), (x1, ww), (y0, hh), (y1, hh)))
                if inside and max(abs(x0 - x1), abs(y0 - y1)) <= 1 and abs(float(cms[b, y0, x0, c]) - float(cms[b, y1, x1, c])) <= map_eps:
                    used[j] = True
                    n_tie += 1
                    continue
            n_only += 1
            assert abs(float(v) - threshold) <= map_eps, f"frame {b}: oracle-only peak with value {v} (channel {c}) at {p}"
        for k in np.where(wi)[0]:  # ill-conditioned refinements: existence (channel + confidence), nearest such peak
            p, v, c = wp[k], wv[k], wc[k]
            same_before, same = same, False
            cand = np.where((gc == c) & ~used & (np.abs(gv - v) <= map_eps))[0]
            if not len(cand) and abs(float(v) - threshold) <= map_eps:
                n_only += 1
                continue
            if not len(cand) and cms is not None and rough_grid is not None:
                x, y = (int(q) for q in rough_grid[np.where(m)[0][k]])
                nb = np.pad(cms[b, :, :, c], 1, constant_values=-np.inf)[y:y + 3, x:x + 3].copy()
                centre = nb[1, 1]
                nb[1, 1] = -np.inf
                if centre - nb.max() <= 2 * map_eps:  # a ridge: not necessarily a strict maximum of the other path's map
                    n_tie += 1
                    continue
            assert len(cand), f"frame {b}: the device has no peak of channel {c} with confidence {v} (ill-conditioned refinement)"
            dd = np.linalg.norm(gp[cand] - p, axis=-1)
            used[cand[int(dd.argmin())]] = True
            if dd.min() <= tol_px:  # (it agrees anyway: not a difference between the two peak sets)
                same = same_before
                if stats is not None:
                    stats["ill_within_tol"] = stats.get("ill_within_tol", 0) + 1
        for j in np.where(~used)[0]:
            same = False
            n_only += 1
            assert abs(float(gv[j]) - threshold) <= map_eps, f"frame {b}: device-only peak with value {gv[j]} (channel {gc[j]})"
        if not same:
            differing.append(b)
            continue
        want = np.asarray(ref[0][b]).reshape(-1, n_nodes, 2)
        got = o["instance_peaks"][b, : len(want)]
        equal = (int(o["n_valid"][b]) == len(want) and np.array_equal(np.isnan(got), np.isnan(want))
                 and (not np.isfinite(got).any() or float(np.nanmax(np.linalg.norm(got - want, axis=-1))) <= tol_px))
        if stats is not None:
            stats.setdefault("frames_with_equal_peak_sets", []).append(b)
            if equal:
                stats.setdefault("frames_with_equal_instances", []).append(b)
        if strict_instances:
            assert int(o["n_valid"][b]) == len(want), f"frame {b}: same peaks, different instance count"
            assert np.array_equal(np.isnan(got), np.isnan(want)), f"frame {b}: same peaks, different node assignment"
            assert equal, f"frame {b}: same peaks, an instance coordinate differs by more than {tol_px} px"
    return differing, n_common, worst, n_only, n_tie


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


def well_conditioned_peaks(cms, refined_grid, rough_grid, vals, sample_inds, channel_inds, threshold=0.2, min_above=0.1, min_margin=0.02,
                           min_ratio=0.9, max_offset=0.6, min_patch_sum=0.8):
    """Which peaks of find_local_peaks(cms, refinement="integral") are decided and refined INSENSITIVELY to a perturbation of the
    maps by the storage precision (a few 1e-3): the detection -- confidence `min_above` over the threshold and the maximum cell
    `min_margin` above its eight neighbours (no neighbouring-cell tie) -- and the refinement -- integral regression is the
    centroid of a 5 x 5 patch, i.e. a division by the patch sum: positive patch (sum >= `min_ratio` of the sum of magnitudes),
    sum >= `min_patch_sum` (a perturbation eps of every cell moves the centroid by at most 25 * 2 eps / sum cells), centroid
    within `max_offset` cells of the centre. On these peaks two paths whose maps agree to ~3e-3 MUST agree within 0.5 px; the
    others are decisions on nearly equal numbers."""
    cms = np.asarray(cms)
    cp = np.pad(cms, ((0, 0), (2, 2), (2, 2), (0, 0)))
    ok = np.zeros(len(rough_grid), bool)
    off = np.abs(np.asarray(refined_grid) - np.asarray(rough_grid)).max(axis=1)
    for k, (p, b, c) in enumerate(zip(rough_grid, sample_inds, channel_inds)):
        x, y = int(p[0]), int(p[1])
        patch = cp[b, y:y + 5, x:x + 5, c]
        ssum, sabs = float(patch.sum()), float(np.abs(patch).sum())
        nb = cp[b, y + 1:y + 4, x + 1:x + 4, c].copy()
        centre = nb[1, 1]
        nb[1, 1] = -np.inf
        ok[k] = (vals[k] >= threshold + min_above and centre - nb.max() >= min_margin and ssum >= min_ratio * sabs
                 and ssum >= min_patch_sum and off[k] <= max_offset)
    return ok
