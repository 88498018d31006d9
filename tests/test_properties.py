"""Property tests (hypothesis) of the invariants SURVEY.md 4.2 lists for the hot path.

CPU part: the oracle restatement itself (so a wrong restatement cannot hide behind a matching
kernel).  -m gpu part: the same properties on the HIP kernels through the C ABI, plus
"cached-feature evaluation == pair evaluation" and batch independence of the towers (F9)."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st
from hypothesis.extra import numpy as hnp

from oracle import kfnet_oracle as O

finite = dict(allow_nan=False, allow_infinity=False, width=32)
coords = hnp.arrays(np.float32, (1, 4, 6, 3), elements=st.floats(-50, 50, **finite))
sigmas = hnp.arrays(np.float32, (1, 4, 6, 1), elements=st.floats(float(np.float32(1e-4)), 50, **finite))


def _kalman_properties(x, s, xm, sm, z, sz):
    """x,s = fused; xm,sm = prediction; z,sz = measurement (fp32)."""
    lo = np.minimum(sm, sz).astype(np.float64)
    # posterior variance: min(P-,R)/2 <= P <= min(P-,R).  In fp32 the gain K = P-/(P-+R) is rounded
    # to 2^-24 before 1-K is formed (KFNet/KFNet.py:154-158), so P = max(1-K,0)*P- carries an
    # ABSOLUTE error of up to ~2^-23 * P- (when R << P- the reference itself returns sigma = 0):
    # the bounds hold with exactly that slack.
    P = s.astype(np.float64) ** 2
    slack = (sm.astype(np.float64) ** 2) * 2.0 ** -22 + 1e-30
    assert np.all(P <= lo * lo * (1 + 1e-5) + slack)
    assert np.all(P >= lo * lo / 2.0 * (1 - 1e-5) - slack)
    # x is a convex combination of prediction and measurement, channel by channel
    a, b = np.minimum(xm, z), np.maximum(xm, z)
    tol = 1e-5 * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))
    assert np.all(x >= a - tol) and np.all(x <= b + tol)


@settings(max_examples=60, deadline=None)
@given(coords, sigmas, coords, sigmas)
def test_oracle_kalman_bounds_and_convexity(xm, sm, z, sz):
    x, s = O.build_kf_coord(xm, sm, z, sz)
    assert x.dtype == np.float32 and s.dtype == np.float32
    _kalman_properties(x, s, xm, sm, z, sz)


@settings(max_examples=40, deadline=None)
@given(coords, sigmas, coords, sigmas)
def test_oracle_nis_is_nonnegative_and_scale_invariant(xm, sm, z, sz):
    n = O.get_nis(z, sz, xm, sm)
    assert np.all(n >= 0)
    # NIS is dimensionless: scaling coordinates and sigmas together leaves it unchanged
    n2 = O.get_nis(z * 4, sz * 4, xm * 4, sm * 4)
    assert np.allclose(n, n2, rtol=2e-4, atol=1e-6)


@settings(max_examples=40, deadline=None)
@given(hnp.arrays(np.float32, (1, 5, 7, 2), elements=st.floats(-3, 9, **finite)),
       hnp.arrays(np.float32, (1, 5, 7, 3), elements=st.floats(-10, 10, **finite)))
def test_oracle_sampler_stays_inside_the_hull_and_zeroes_outside(pm, img):
    """tools/util.py:36-93: interior samples are convex combinations of the 4 neighbours; any
    sample with x < 0, x >= W-1, y < 0 or y >= H-1 is exactly 0 (clamped-weight rule, App. A6)."""
    out = O.bilinear_sampler(img, pm)
    H, W = 5, 7
    x, y = pm[0, ..., 0], pm[0, ..., 1]
    outside = (x < 0) | (x >= W - 1) | (y < 0) | (y >= H - 1)
    assert np.all(np.abs(out[0][outside]) <= 1e-4)
    assert np.all(out <= max(img.max(), 0.0) + 1e-4) and np.all(out >= min(img.min(), 0.0) - 1e-4)


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.integers(0, 2 ** 31 - 1))
def test_oracle_towers_are_batch_independent(seed):
    """no layer couples batch elements, so evaluating a frame alone or inside a
    pair batch gives the same maps -- what lets the native path run the towers once per frame."""
    from kfnet_amd.weights import synthetic_weights
    W = test_oracle_towers_are_batch_independent.W = getattr(test_oracle_towers_are_batch_independent, 'W', None) \
        or synthetic_weights(3)
    rng = np.random.default_rng(seed)
    imgs = rng.integers(0, 256, size=(2, 16, 24, 3), dtype=np.uint8)
    f_pair = O.oflow_feat(imgs, W, np.float32)
    f_one = O.oflow_feat(imgs[1:2], W, np.float32)
    assert np.allclose(f_pair[1:2], f_one, atol=1e-6)


# ---- the same properties on the HIP kernels --------------------------------------------------
@pytest.mark.gpu
@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 5000))
def test_hip_kalman_fuse_bounds_and_convexity(seed, P):
    import torch
    from kfnet_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    pred = rng.uniform(-50, 50, size=(P, 4)).astype(np.float32)
    meas = rng.uniform(-50, 50, size=(P, 4)).astype(np.float32)
    pred[:, 3] = np.exp(rng.uniform(np.log(1e-4), np.log(50), size=P)).astype(np.float32)
    meas[:, 3] = np.exp(rng.uniform(np.log(1e-4), np.log(50), size=P)).astype(np.float32)
    dp, dm = torch.from_numpy(pred).cuda(), torch.from_numpy(meas).cuda()
    out = torch.empty_like(dp)
    _lib.check(lib.kfn_kalman_fuse(dp.data_ptr(), dm.data_ptr(), out.data_ptr(), None, P,
                                   torch.cuda.current_stream().cuda_stream), 'kfn_kalman_fuse')
    o = out.cpu().numpy()
    _kalman_properties(o[:, :3], o[:, 3:4], pred[:, :3], pred[:, 3:4], meas[:, :3], meas[:, 3:4])
    # and bit-exact against the fp32 oracle (KFNet/KFNet.py:148-162 in its operation order)
    rx, rs = O.build_kf_coord(pred[None, None, :, :3], pred[None, None, :, 3:4], meas[None, None, :, :3],
                              meas[None, None, :, 3:4])
    assert np.array_equal(o[:, :3], rx[0, 0]) and np.array_equal(o[:, 3:4], rs[0, 0])


@pytest.mark.gpu
@settings(max_examples=4, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.sampled_from([1, 2, 3]))
def test_hip_cached_feature_path_equals_pair_evaluation(seed, batch):
    """The engine keeps frame t-1's flow features in a ring and runs every tower once per frame;
    the reference evaluates every (t-1, t) pair from scratch (KFNet/eval.py:41, batch 2).  Flow
    and transition sigma of frame t must not depend on which of the two was used -- nor on the
    tower batch size (F9)."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(17)
    imgs = synthetic_sequence(5, 64, 96, seed=seed % 1000)
    eng = KFNetEngine(W, image_size=(64, 96), batch=batch, reset_period=500, max_chunk=8)
    eng.process(eng.upload_frames(imgs))
    seq = eng.debug(5)
    for t in (1, 3, 4):
        pair = KFNetEngine(W, image_size=(64, 96), batch=2, reset_period=500, max_chunk=4)
        pair.process(pair.upload_frames(imgs[t - 1:t + 1]))      # (t-1, t) from scratch
        d = pair.debug(2)
        assert np.allclose(seq['flow'][t], d['flow'][1], atol=2e-5)
        assert np.allclose(seq['sigma_trans'][t], d['sigma_trans'][1], rtol=1e-4)
        assert np.allclose(seq['meas'][t], d['meas'][1], atol=2e-5, rtol=1e-4)


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 2), st.integers(1, 4), st.integers(1, 5), st.integers(0, 2 ** 31 - 1))
def test_polyphase_weight_fragments_reproduce_the_stride2_convolution(n, th, tw, seed):
    """For any even image size: the 16 pre-signed fragments of pack_winograd_s2_kernel, applied per phase with the
    F(2,2) output transform written out explicitly (no accumulator folding here -- that is checked in
    test_oracle_kat), give the oracle's stride-2 SAME convolution."""
    from kfnet_amd.graph import pack_winograd_s2_kernel
    rng = np.random.default_rng(seed)
    H, W, ci, co = 4 * th - 2 * int(rng.integers(0, 2)), 4 * tw - 2 * int(rng.integers(0, 2)), 8, 3
    x = rng.normal(size=(n, H, W, ci))
    wt = rng.normal(size=(3, 3, ci, co)).astype(np.float32)
    ref = O.conv2d_same(x, wt.astype(np.float64), None, 2, False)
    u = pack_winograd_s2_kernel(wt)
    U = u.transpose(1, 0, 3, 2).reshape(16, ci, u.shape[2])[:, :, :co].astype(np.float64)
    # the packer's minus sign on Winograd index 2 turns A^T = [[1,1,0],[0,1,-1]] into [[1,1,0],[0,1,1]]
    AT = np.array([[1.0, 1.0, 0.0], [0.0, 1.0, 1.0]])
    BT = np.array([[1.0, -1.0, 0.0], [0.0, 1.0, 0.0], [0.0, 1.0, -1.0]])
    xp = np.pad(x, ((0, 0), (0, 4), (0, 4), (0, 0)))
    got = np.zeros_like(ref)
    for im in range(n):
        for ty in range((H // 2 + 1) // 2):
            for tx in range((W // 2 + 1) // 2):
                d = xp[im, 4 * ty:4 * ty + 5, 4 * tx:4 * tx + 5]
                v00 = np.einsum('xm,mnc,yn->xyc', BT, d[0::2, 0::2], BT)
                m00 = np.einsum('xyc,xyco->xyo', v00, U[0:9].reshape(3, 3, ci, co))
                y = np.einsum('ax,xyo,by->abo', AT, m00, AT)
                v01 = np.einsum('xm,mnc->xnc', BT, d[0::2, 1::2])                  # [3,2,ci]
                y += np.einsum('ax,xno->ano', AT, np.einsum('xnc,xco->xno', v01, U[9:12]))
                v10 = np.einsum('yn,mnc->myc', BT, d[1::2, 0::2])                  # [2,3,ci]
                y += np.einsum('by,myo->mbo', AT, np.einsum('myc,yco->myo', v10, U[12:15]))
                y += np.einsum('mnc,co->mno', d[1::2, 1::2], U[15])
                for di in range(2):
                    for dj in range(2):
                        oy, ox = 2 * ty + di, 2 * tx + dj
                        if oy < H // 2 and ox < W // 2:
                            got[im, oy, ox] = y[di, dj]
    assert np.abs(got - ref).max() < 1e-5
