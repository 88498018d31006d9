"""Discontinuity-aware comparison of two runs of the KFNet prediction path.

The reference's bilinear sampler (tools/util.py:36-93) clamps the corner INDICES and computes
the weights FROM the clamped corners; a warped sample with x < 0 or x >= W-1 (resp. y) therefore
evaluates to exactly 0 while a sample just inside evaluates to the border pixel's value: the
process model x^- = bilinear(last_coord, pixel_map + flow) (KFNet/KFNet.py:381-403) is a step
function of the flow at x in {0, W-1} and y in {0, H-1}.  Two arithmetically different
evaluations of the same network (fp32 vs fp16-operand convolutions) whose flows differ by 1e-3 px
can land on different sides of a step; the affected pixel then differs by the full value of the
state (not by round-off), keeps that difference in the recurrent state and hands it on to every
pixel that later samples from it.

`sampler_taint` marks the pixels such an event can reach:
  * frame t, pixel p is a SEED if the two runs' sample positions lie on DIFFERENT sides of a step
    (`crossing_seeds`, needs both flows) -- or, from the reference run's flow alone, if its sample
    position lies within `delta` px of one of the four steps (|x| < delta, |x - (W-1)| < delta, same
    in y; conservative: every pixel of a border row/column with a small flow is such a seed);
  * the taint propagates exactly as the state does: p is tainted at t if it is a seed at t or its
    sample at t lies inside the grid and one of the (up to) four corner pixels it reads was tainted
    at t-1; a reset frame (t % reset_period == 0: the state is overwritten by the measurement,
    KFNet/eval.py:94-101) clears everything.
`masked_parity` then states the tolerance on the un-tainted pixels and reports the tainted
fraction; nothing is excluded silently.
"""
import numpy as np


def sample_positions(flow):
    """pixel_map + flow (KFNet/util.py:42-63 + KFNet/KFNet.py:386): flow [T,h,w,2] as (x,y)."""
    T, h, w, _ = flow.shape
    xs = np.arange(w, dtype=np.float32)[None, None, :] + flow[..., 0]
    ys = np.arange(h, dtype=np.float32)[None, :, None] + flow[..., 1]
    return xs, ys


def inside_grid(flow):
    """bool [T,h,w]: does the sample land where the reference sampler returns data (0 <= x < W-1, 0 <= y < H-1)?"""
    T, h, w, _ = flow.shape
    xs, ys = sample_positions(np.asarray(flow, dtype=np.float32))
    return (xs >= 0) & (xs < w - 1) & (ys >= 0) & (ys < h - 1)


def step_distance(flow):
    """Distance [T,h,w] (px) of every sample position from the nearest of the sampler's four steps."""
    T, h, w, _ = flow.shape
    xs, ys = sample_positions(np.asarray(flow, dtype=np.float32))
    return np.minimum.reduce([np.abs(xs), np.abs(xs - (w - 1)), np.abs(ys), np.abs(ys - (h - 1))])


def crossing_seeds(ref_flow, test_flow, reset_period=500, t0=0):
    """bool [T,h,w]: the two runs' samples fall on different sides of a sampler step (one reads the state, the
    other gets 0).  Reset frames sample nothing."""
    seeds = inside_grid(ref_flow) != inside_grid(test_flow)
    for t in range(seeds.shape[0]):
        if reset_period > 0 and (t0 + t) % reset_period == 0:
            seeds[t] = False
    return seeds


def sampler_taint(flow, delta=0.05, reset_period=500, t0=0, seeds=None):
    """bool [T,h,w]: pixels whose state can differ by more than round-off between two evaluations
    (see module docstring).  `flow` is the reference run's [T,h,w,2]; frame t of the array is global frame
    t0 + t.  `seeds` = crossing_seeds(...) when both flows are known, else the `delta` neighbourhood of the steps."""
    flow = np.asarray(flow, dtype=np.float32)
    T, h, w, _ = flow.shape
    xs, ys = sample_positions(flow)
    seed = seeds if seeds is not None else (step_distance(flow) < delta)
    inside = (xs >= 0) & (xs < w - 1) & (ys >= 0) & (ys < h - 1)   # elsewhere the sample is 0: reads no state
    x0 = np.clip(np.floor(xs).astype(np.int64), 0, w - 1)
    y0 = np.clip(np.floor(ys).astype(np.int64), 0, h - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    y1 = np.clip(y0 + 1, 0, h - 1)
    taint = np.zeros((T, h, w), dtype=bool)
    prev = np.zeros((h, w), dtype=bool)
    for t in range(T):
        if reset_period > 0 and (t0 + t) % reset_period == 0:
            prev = np.zeros((h, w), dtype=bool)        # state := measurement, nothing is sampled
        else:
            reads = prev[y0[t], x0[t]] | prev[y0[t], x1[t]] | prev[y1[t], x0[t]] | prev[y1[t], x1[t]]
            prev = seed[t] | (inside[t] & reads)
        taint[t] = prev
    return taint


def masked_parity(rec, ref, ref_flow, coord_tol=2e-2, conf_rel_tol=5e-2, delta=0.05, reset_period=500, t0=0,
                  test_flow=None):
    """Compare records `rec` with `ref` ([T,h,w,4]: T.x and 1/sigma, KFNet/eval.py:115,123-126)
    away from the sampler's steps.  With `test_flow` (the compared run's own flow) the mask is seeded by the
    ACTUAL step crossings and `delta` becomes a checked claim (every crossing lies within delta of a step and the
    flows agree to better than delta); without it, by the delta neighbourhood.  Returns a dict of plain floats."""
    rec, ref = np.asarray(rec), np.asarray(ref)
    seeds = None
    extra = {}
    if test_flow is not None:
        seeds = crossing_seeds(ref_flow, test_flow, reset_period, t0)
        sd = step_distance(ref_flow)
        # frames that sample something: on a reset frame the flow is computed but never used (and pairs the frame
        # with whatever the feature ring held before the sequence started)
        used = np.array([not (reset_period > 0 and (t0 + t) % reset_period == 0) for t in range(rec.shape[0])])
        fd = np.abs(np.asarray(ref_flow) - np.asarray(test_flow))[used]
        extra = {'crossings': int(seeds.sum()),
                 'crossing_max_step_distance_px': float(sd[seeds].max()) if seeds.any() else 0.0,
                 'flow_max_abs_diff_px': float(fd.max()) if fd.size else 0.0}
    taint = sampler_taint(ref_flow, delta, reset_period, t0, seeds=seeds)
    dc = np.abs(rec[..., :3] - ref[..., :3]).max(-1)
    dr = np.abs(rec[..., 3] - ref[..., 3]) / np.abs(ref[..., 3])
    bad = (dc > coord_tol) | (dr > conf_rel_tol)
    clean = ~taint
    return {
        'frames': int(rec.shape[0]), 'pixels': int(dc.size), 'delta_px': float(delta),
        'coord_tol': float(coord_tol), 'conf_rel_tol': float(conf_rel_tol),
        'masked_fraction': float(taint.mean()),
        'unmasked_outside_tolerance': int((bad & clean).sum()),
        'unmasked_coord_max_abs': float(dc[clean].max()) if clean.any() else 0.0,
        'unmasked_conf_max_rel': float(dr[clean].max()) if clean.any() else 0.0,
        'masked_outside_tolerance': int((bad & taint).sum()),
        'all_pixels_coord_max_abs': float(dc.max()), 'all_pixels_conf_max_rel': float(dr.max()),
        'coord_abs_p999': float(np.quantile(dc, 0.999)), 'conf_rel_p999': float(np.quantile(dr, 0.999)),
        'outside_tolerance_fraction': float(bad.mean()),
        'mask': 'descendants of actual step crossings' if seeds is not None else 'descendants of the delta neighbourhood',
        **extra,
    }


def merge_parity(parts):
    """Combine `masked_parity` results of several independent sequences."""
    n = float(sum(p['pixels'] for p in parts))
    out = dict(parts[0])
    out['frames'] = int(sum(p['frames'] for p in parts))
    out['pixels'] = int(n)
    out['masked_fraction'] = float(sum(p['masked_fraction'] * p['pixels'] for p in parts) / n)
    for k in ('unmasked_outside_tolerance', 'masked_outside_tolerance', 'crossings'):
        if k in parts[0]:
            out[k] = int(sum(p[k] for p in parts))
    out['outside_tolerance_fraction'] = float(sum(p['outside_tolerance_fraction'] * p['pixels'] for p in parts) / n)
    for k in ('crossing_max_step_distance_px', 'flow_max_abs_diff_px'):
        if k in parts[0]:
            out[k] = float(max(p[k] for p in parts))
    for k in ('unmasked_coord_max_abs', 'unmasked_conf_max_rel', 'all_pixels_coord_max_abs',
              'all_pixels_conf_max_rel', 'coord_abs_p999', 'conf_rel_p999'):
        out[k] = float(max(p[k] for p in parts))
    return out
