"""kfnet_amd/tools/parity.py: the discontinuity mask used to state config 5's tolerance.

Checked against the fp32 numpy oracle of the scan (oracle/kfnet_oracle.py: bilinear_sampler follows
tools/util.py:36-93): two scans whose flows differ by <= 1e-3 px may differ by O(1) at pixels whose
sample crosses the sampler's step at x in {0, W-1} / y in {0, H-1} -- and at pixels that later read
those -- but nowhere else; every large difference must be inside `sampler_taint`."""
import numpy as np

from kfnet_amd.tools.parity import masked_parity, sampler_taint
from test_dist_gloo import _inputs, _scan


def test_sampler_step_is_where_the_oracle_says():
    from oracle import kfnet_oracle as O
    img = np.arange(1, 1 + 5 * 7, dtype=np.float32).reshape(1, 5, 7, 1)
    def at(x, y):
        pm = np.zeros((1, 5, 7, 2), np.float32)
        pm[..., 0], pm[..., 1] = x, y
        return float(O.bilinear_sampler(img, pm)[0, 0, 0, 0])
    assert at(-1e-3, 2.0) == 0.0 and at(0.0, 2.0) == img[0, 2, 0, 0]          # step at x = 0 (from the left)
    assert at(6.0, 2.0) == 0.0 and abs(at(6.0 - 1e-3, 2.0) - img[0, 2, 6, 0]) < 2e-3   # step at x = W-1
    assert at(3.0, -1e-3) == 0.0 and at(3.0, 4.0) == 0.0 and at(3.0, 4.0 - 1e-3) > 0   # same in y


def test_large_differences_are_confined_to_the_taint():
    T, H, W = 24, 12, 17
    flow, sig, meas = _inputs(T, H, W, seed=5)
    rng = np.random.default_rng(9)
    pert = rng.uniform(-1e-3, 1e-3, size=flow.shape).astype(np.float32)
    # plant a few samples right beside a step, with a perturbation that crosses it
    for (t, y, x, edge) in [(3, 4, 5, 'x0'), (7, 8, 2, 'xW'), (13, 1, 9, 'y0'), (17, 6, 11, 'yH'), (22, 3, 3, 'x0')]:
        if edge[0] == 'x':
            flow[t, y, x, 0] = (0.0 if edge == 'x0' else W - 1.0) - x + (3e-4 if edge == 'x0' else -3e-4)
            pert[t, y, x, 0] = -8e-4 if edge == 'x0' else 8e-4
            flow[t, y, x, 1] = 0.3
        else:
            flow[t, y, x, 1] = (0.0 if edge == 'y0' else H - 1.0) - y + (3e-4 if edge == 'y0' else -3e-4)
            pert[t, y, x, 1] = -8e-4 if edge == 'y0' else 8e-4
            flow[t, y, x, 0] = 0.3
    flow2 = flow + pert
    z = np.zeros((H, W, 4), np.float32)
    a, _ = _scan(flow, sig, meas, z, 0, 10)
    b, _ = _scan(flow2, sig, meas, z, 0, 10)
    taint = sampler_taint(flow, delta=2e-3, reset_period=10)
    d = np.abs(a[..., :3] - b[..., :3]).max(-1)
    assert (d > 0.1).any(), 'the case must contain at least one step crossing'
    assert not (d[~taint] > 2e-2).any()
    assert taint[0].sum() == 0 and taint[10].sum() == 0          # reset frames clear it
    r = masked_parity(b, a, flow, coord_tol=2e-2, conf_rel_tol=1e9, delta=2e-3, reset_period=10)
    assert r['unmasked_outside_tolerance'] == 0 and r['masked_outside_tolerance'] > 0
    assert 0 < r['masked_fraction'] < 0.5


def test_taint_propagates_along_the_flow():
    T, H, W = 4, 6, 8
    flow = np.zeros((T, H, W, 2), np.float32)
    flow[1, 2, 0, 0] = -0.01                 # frame 1: pixel (2,0) samples at x = -0.01 -> seed
    flow[2, 3, 4, :] = (-3.5, -0.5)          # frame 2: pixel (3,4) samples at (0.5, 2.5) -> reads (2,0)
    t = sampler_taint(flow, delta=0.05, reset_period=500)
    # x = 0 exactly (zero flow in column 0) sits ON the step: also seeds
    assert t[1, 2, 0] and t[2, 3, 4] and not t[2, 3, 5] and not t[0].any()


def test_crossing_seeds_explain_every_large_difference():
    """With BOTH flows known the mask is seeded by the actual crossings (samples on different sides of a step):
    far fewer pixels than the delta neighbourhood, and still every large difference is a descendant."""
    from kfnet_amd.tools.parity import crossing_seeds
    T, H, W = 24, 12, 17
    flow, sig, meas = _inputs(T, H, W, seed=5)
    rng = np.random.default_rng(9)
    pert = rng.uniform(-1e-3, 1e-3, size=flow.shape).astype(np.float32)
    for (t, y, x) in [(3, 4, 5), (13, 1, 9), (22, 3, 3)]:
        flow[t, y, x, 0] = -x + 3e-4
        pert[t, y, x, 0] = -8e-4
        flow[t, y, x, 1] = 0.3
    flow2 = flow + pert
    z = np.zeros((H, W, 4), np.float32)
    a, _ = _scan(flow, sig, meas, z, 0, 10)
    b, _ = _scan(flow2, sig, meas, z, 0, 10)
    seeds = crossing_seeds(flow, flow2, reset_period=10)
    assert seeds.sum() >= 3 and seeds[3, 4, 5] and seeds[13, 1, 9] and seeds[22, 3, 3]
    r = masked_parity(b, a, flow, coord_tol=2e-2, conf_rel_tol=1e9, delta=2e-3, reset_period=10, test_flow=flow2)
    r_delta = masked_parity(b, a, flow, coord_tol=2e-2, conf_rel_tol=1e9, delta=2e-3, reset_period=10)
    assert r['unmasked_outside_tolerance'] == 0 and r['masked_outside_tolerance'] > 0
    assert r['masked_fraction'] <= r_delta['masked_fraction']
    assert r['crossing_max_step_distance_px'] < 2e-3 and r['flow_max_abs_diff_px'] <= 1.01e-3
