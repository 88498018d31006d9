"""-m gpu parity tests: every C-ABI entry point of libkfnet_hip.so against the CPU oracle
on identical seeded inputs.  fp32 tolerance of the convolution kernels: the error model of tests/conv_tol.py
(a multiple of eps32 * sqrt(K) * rms(x) * rms(w) * transform gain; every check logs measured error / bound)."""
import ctypes as C

import numpy as np
import pytest

from oracle import kfnet_oracle as O

pytestmark = pytest.mark.gpu


from tests.conv_tol import assert_close, conv_err_bound, record  # noqa: E402


def _conv_tol(x, w, transposed=False, kind='direct'):
    return conv_err_bound(x, w, kind, transposed)


def _check_err(err, x, w, kind, label, transposed=False):
    """Scalar form: `err` (already reduced by the caller) against the model's bound; logged like assert_close."""
    bound = conv_err_bound(x, w, kind, transposed)
    record(kind, label, float(err), bound)
    assert err <= bound, '%s %s: max err %.3e, bound %.3e' % (kind, label, err, bound)


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, relu
    (1, 8, 8, 32, 32, 3, 1, True),
    (2, 13, 17, 16, 48, 3, 1, False),
    (1, 12, 20, 64, 64, 3, 2, True),       # even size stride 2: pad (0,1)
    (1, 15, 9, 32, 100, 3, 2, True),       # odd size stride 2: pad (1,1)
    (3, 8, 8, 48, 16, 3, 1, True),         # OFlowNet conv6 shape class
    (5, 8, 8, 16, 1, 3, 1, False),         # OFlowNet prediction: Cout = 1
    (2, 6, 10, 256, 128, 1, 1, True),      # 1x1
    (1, 20, 24, 128, 320, 3, 1, True),     # multiple N tiles
    (7, 1, 1, 128, 128, 3, 1, True),       # 1x1 spatial: only the centre tap is live
    (6, 2, 2, 64, 64, 3, 1, True),
    (1, 60, 80, 64, 160, 3, 1, True),      # M = 4800 = 30 x 160
]


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('config', [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
def test_conv_vs_oracle(case, config):
    from tests.gpu_util import run_conv
    n, h, w, ci, co, k, s, relu = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.normal(size=(n, h, w, ci)).astype(np.float32)
    wt = (rng.normal(size=(k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    y = run_conv(x, wt, b, s, relu, config=config)
    ref = O.conv2d_same(x.astype(np.float64), wt, b, s, relu)
    assert y.shape == ref.shape
    _check_err(np.abs(y - ref).max(), x, wt, 'direct', 'conv %s cfg %d' % (case, config))


def test_conv_strided_views():
    from tests.gpu_util import run_conv
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2, 8, 8, 32)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, 32, 16)) / 17).astype(np.float32)
    b = rng.normal(size=16).astype(np.float32)
    y = run_conv(x, wt, b, 1, True, ldx=96, x_off=64, ldy=48, y_off=16)
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 1, True)
    _check_err(np.abs(y - ref).max(), x, wt, 'direct', 'conv strided views')


@pytest.mark.parametrize('shape', [(3, 1, 1, 128, 64), (2, 2, 2, 64, 32), (2, 4, 4, 32, 16), (1, 5, 7, 16, 40),
                                   (300, 2, 2, 64, 32), (70, 4, 4, 32, 16)])
@pytest.mark.parametrize('config', [0, 4, 10, 11])
def test_deconv_vs_oracle(shape, config):
    from tests.gpu_util import run_conv
    n, h, w, ci, co = shape
    rng = np.random.default_rng(11)
    x = rng.normal(size=(n, h, w, ci)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, co, ci)) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    y = run_conv(x, wt, b, 2, True, transposed=True, config=config)
    ref = O.conv2d_transpose_same(x.astype(np.float64), wt, b, 2, True)
    assert y.shape == ref.shape
    _check_err(np.abs(y - ref).max(), x, wt, 'direct', 'deconv %s cfg %d' % (shape, config), transposed=True)


def test_conv_epilogues():
    from tests.gpu_util import run_conv
    from kfnet_amd import _lib
    rng = np.random.default_rng(21)
    x = rng.normal(size=(2, 6, 7, 128)).astype(np.float32)
    w32 = (rng.normal(size=(3, 3, 128, 32)) / 34).astype(np.float32)
    b32 = rng.normal(size=32).astype(np.float32)
    y = run_conv(x, w32, b32, 1, False, epilogue=_lib.EPI_L2NORM)
    ref = O.l2_normalize(O.conv2d_same(x.astype(np.float64), w32, b32, 1, False))
    assert np.abs(y - ref).max() < 2e-6
    w4 = (rng.normal(size=(1, 1, 128, 4)) / 11).astype(np.float32)
    b4 = rng.normal(size=4).astype(np.float32)
    y = run_conv(x, w4, b4, 1, False, epilogue=_lib.EPI_EXP_CH3)
    ref = O.conv2d_same(x.astype(np.float64), w4, b4, 1, False)
    ref[..., 3] = np.exp(ref[..., 3])
    assert np.abs(y - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    w1 = (rng.normal(size=(1, 1, 128, 1)) / 11).astype(np.float32)
    y = run_conv(x, w1, None, 1, False, epilogue=_lib.EPI_EXP_1E2)
    ref = np.exp(O.conv2d_same(x.astype(np.float64), w1, None, 1, False)) * 1e-2
    assert np.abs(y - ref).max() < 1e-6


def test_conv_bad_args_fail_loudly():
    from tests.gpu_util import dev, stream
    from kfnet_amd import _lib
    lib = _lib.load()
    x = dev(np.zeros((1, 4, 4, 8), np.float32))
    d = _lib.ConvDesc(N=1, H=4, W=4, Cin=8, ldx=8, Cout=8, cout_pad=32, ldy=8, kh=3, kw=3, stride=1)
    rc = lib.kfn_conv2d_nhwc(C.byref(d), x.data_ptr(), x.data_ptr(), None, x.data_ptr(), stream())
    assert rc == -1 and b'Cin' in lib.kfn_last_error()


@pytest.mark.parametrize('hw', [(8, 64), (9, 70), (48, 96), (37, 132), (5, 4), (540 // 8, 960)])
def test_first_conv_u8(hw):
    """W*3 % 4 == 0 takes the lane-per-pixel kernel (aligned dword halo loads; (37,132): a partial last tile in x and
    y, (5,4): an image narrower than the halo), the other widths the byte-wise kernel; a single-head call and guard
    words behind both outputs."""
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    import torch
    lib = _lib.load()
    H, W = hw
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(2, H, W, 3), dtype=np.uint8)
    w1 = (rng.normal(size=(3, 3, 3, 64)) / 5).astype(np.float32)
    b1 = rng.normal(size=64).astype(np.float32)
    w2 = (rng.normal(size=(3, 3, 3, 16)) / 5).astype(np.float32)
    b2 = rng.normal(size=16).astype(np.float32)
    n1, n2 = 2 * H * W * 64, 2 * H * W * 16
    y1 = torch.full((n1 + 256,), -3.0, device='cuda')
    y2 = torch.full((n2 + 256,), -3.0, device='cuda')
    di, dw1, db1, dw2, db2 = dev(img), dev(w1.reshape(27, 64)), dev(b1), dev(w2.reshape(27, 16)), dev(b2)
    _lib.check(lib.kfn_first_conv_u8(di.data_ptr(), 2, H, W, dw1.data_ptr(), db1.data_ptr(), y1.data_ptr(), 64,
                                     dw2.data_ptr(), db2.data_ptr(), y2.data_ptr(), 16, stream()), 'first')
    sync()
    xp = O.preprocess(img, np.float64)
    r1 = O.conv2d_same(xp, w1, b1, 1, True)
    r2 = O.conv2d_same(xp, w2, b2, 1, True)
    g1, g2 = y1.cpu().numpy(), y2.cpu().numpy()
    assert np.all(g1[n1:] == -3.0) and np.all(g2[n2:] == -3.0)
    assert np.abs(g1[:n1].reshape(r1.shape) - r1).max() < 2e-5
    assert np.abs(g2[:n2].reshape(r2.shape) - r2).max() < 2e-5
    # one head only, no bias
    y1b = torch.full((n1 + 256,), -3.0, device='cuda')
    _lib.check(lib.kfn_first_conv_u8(di.data_ptr(), 2, H, W, dw1.data_ptr(), None, y1b.data_ptr(), 64, None, None, None,
                                     0, stream()), 'first (one head)')
    sync()
    g1b = y1b.cpu().numpy()
    assert np.all(g1b[n1:] == -3.0)
    assert np.abs(g1b[:n1].reshape(r1.shape) - O.conv2d_same(xp, w1, None, 1, True)).max() < 2e-5


def test_cost_volume_bit_exact():
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(4)
    N, H, W, Cc = 3, 7, 9, 32
    f = rng.normal(size=(N + 1, H, W, Cc)).astype(np.float32)
    fd = dev(f)
    vol = torch.zeros(N * H * W * 64 * Cc, device='cuda')
    _lib.check(lib.kfn_cost_volume(fd.data_ptr(), fd.data_ptr() + H * W * Cc * 4, vol.data_ptr(), N, H, W, Cc, 8,
                                   stream()), 'cv')
    sync()
    got = vol.cpu().numpy().reshape(N, H * W, 8, 8, Cc)
    for n in range(N):
        ref, _ = O.coord_volume(f[n:n + 1], f[n + 1:n + 2], 8)
        assert np.array_equal(got[n], ref)   # subtraction only: bit exact


def test_flow_softargmax():
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(6)
    P = 1003
    logits = (rng.normal(size=(P, 64)) * 4).astype(np.float32)
    flow = torch.zeros(P * 2, device='cuda')
    prob = torch.zeros(P * 64, device='cuda')
    dl = dev(logits)
    _lib.check(lib.kfn_flow_softargmax(dl.data_ptr(), flow.data_ptr(), prob.data_ptr(), P, 8, stream()), 'flow')
    sync()
    pr = O.softmax(logits.astype(np.float64))
    offs = O.coord_volume(np.zeros((1, 2, 2, 1)), np.zeros((1, 2, 2, 1)), 8)[1]
    assert np.abs(prob.cpu().numpy().reshape(P, 64) - pr).max() < 1e-6
    assert np.abs(flow.cpu().numpy().reshape(P, 2) - pr.dot(offs)).max() < 5e-6
    # uniform logits -> (-0.5, -0.5) exactly
    dl = dev(np.zeros((4, 64), np.float32))
    _lib.check(lib.kfn_flow_softargmax(dl.data_ptr(), flow.data_ptr(), None, 4, 8, stream()), 'flow')
    sync()
    assert np.all(flow.cpu().numpy()[:8] == -0.5)


def _scan_ref(flow, sig, meas, state0, T4, t0, reset_period, nis_gate):
    """fp32 numpy oracle of the scan, frame by frame."""
    S, T, H, W, _ = flow.shape
    offs = None
    rec = np.zeros((S, T, H, W, 4), np.float32)
    temp = np.zeros((S, T, H, W, 4), np.float32)
    nis = np.zeros((S, T, H, W, 3), np.float32)
    state = state0.copy()
    for s in range(S):
        sx, ss = state[s:s + 1, ..., 0:3], state[s:s + 1, ..., 3:4]
        for t in range(T):
            z, sz = meas[s, t][None, ..., 0:3], meas[s, t][None, ..., 3:4]
            if reset_period > 0 and (t0 + t) % reset_period == 0:
                ox, os_ = z, sz
                sx, ss = z, sz
                temp[s, t] = meas[s, t]
            else:
                pm = O.get_pixel_map(H, W, np.float32) + flow[s, t][None]
                tx = O.bilinear_sampler(sx, pm)
                lu = O.bilinear_sampler(ss, pm)
                eps2 = np.float32(1e-5) * np.float32(1e-5)
                ts = np.sqrt(np.maximum(sig[s, t][None] ** 2, eps2) + np.maximum(lu * lu, eps2))
                kx, ks = O.build_kf_coord(tx, ts, z, sz)
                nn = O.get_nis(z, sz, tx, ts)
                ox, os_ = kx, ks
                if nis_gate > 0:
                    m = ((nn[..., 0:1] + nn[..., 1:2]) + nn[..., 2:3]) > nis_gate
                    ox = np.where(m, z, kx)
                sx, ss = kx, ks
                temp[s, t] = np.concatenate([tx[0], ts[0]], -1)
                nis[s, t] = nn[0]
            if T4 is not None:
                ox = O.apply_transform(ox, T4)
            rec[s, t] = np.concatenate([ox[0], 1.0 / os_[0]], -1)
        state[s] = np.concatenate([sx[0], ss[0]], -1)
    return rec, temp, nis, state


@pytest.mark.parametrize('cfg', [(1, 3, 60, 80, 0, 500, 0.0, True), (3, 5, 17, 23, 498, 500, 7.815, True),
                                 (2, 4, 68, 120, 1, 0, 0.0, False),
                                 # state larger than the LDS (> 10 240 px): per-frame kernel, global ping-pong
                                 (2, 5, 101, 121, 2, 4, 7.815, True), (1, 4, 135, 240, 0, 500, 0.0, False)])
def test_kalman_scan_vs_oracle(cfg):
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.synth import synthetic_transform
    import torch
    lib = _lib.load()
    S, T, H, W, t0, rp, gate, use_T = cfg
    rng = np.random.default_rng(S * 100 + T)
    flow = (rng.normal(size=(S, T, H, W, 2)) * 2.0).astype(np.float32)
    flow[:, :, 0, :, :] = -3.0  # push a row out of range
    sig = np.abs(rng.normal(size=(S, T, H, W, 1)) * 0.05).astype(np.float32)
    meas = rng.normal(size=(S, T, H, W, 4)).astype(np.float32)
    meas[..., 3] = np.abs(meas[..., 3]) * 0.3 + 0.05
    state0 = rng.normal(size=(S, H, W, 4)).astype(np.float32)
    state0[..., 3] = np.abs(state0[..., 3]) * 0.3 + 0.05
    T4 = O.get_transform(synthetic_transform()) if use_T else None
    d = _lib.KalmanDesc(S=S, T=T, H=H, W=W, t0=t0, reset_period=rp, min_uncertainty=1e-5, nis_gate=gate,
                        has_transform=int(use_T))
    if use_T:
        for i, v in enumerate(T4[:3].reshape(-1)):
            d.transform[i] = float(v)
    dfl, dsg, dme, dst = dev(flow), dev(sig), dev(meas), dev(state0)
    rec = torch.zeros(S * T * H * W * 4, device='cuda')
    tmp = torch.zeros(S * T * H * W * 4, device='cuda')
    nis = torch.zeros(S * T * H * W * 3, device='cuda')
    need = C.c_size_t(99)
    _lib.check(lib.kfn_kalman_scan_scratch_bytes(C.byref(d), C.byref(need)), 'scratch bytes')
    assert need.value == (S * H * W * 16 if H * W > 10240 else 0)      # a second copy of the state iff it cannot live in LDS
    scratch = torch.empty(need.value // 4, device='cuda') if need.value else None
    if need.value:     # the library allocates nothing: without the caller's scratch a large grid is an argument error
        assert lib.kfn_kalman_scan(C.byref(d), dfl.data_ptr(), dsg.data_ptr(), dme.data_ptr(), dst.data_ptr(),
                                   rec.data_ptr(), tmp.data_ptr(), nis.data_ptr(), None, stream()) == -1
        assert b'scratch' in lib.kfn_last_error()
    _lib.check(lib.kfn_kalman_scan(C.byref(d), dfl.data_ptr(), dsg.data_ptr(), dme.data_ptr(), dst.data_ptr(),
                                   rec.data_ptr(), tmp.data_ptr(), nis.data_ptr(),
                                   scratch.data_ptr() if scratch is not None else None, stream()), 'scan')
    sync()
    r_rec, r_tmp, r_nis, r_state = _scan_ref(flow, sig, meas, state0, T4, t0, rp, gate)
    g_rec = rec.cpu().numpy().reshape(r_rec.shape)
    g_state = dst.cpu().numpy().reshape(r_state.shape)
    # elementwise fp32 with the same op order: a few ulps (transform matmul order aside)
    assert np.allclose(tmp.cpu().numpy().reshape(r_tmp.shape), r_tmp, rtol=2e-6, atol=2e-6)
    assert np.allclose(g_state, r_state, rtol=4e-6, atol=4e-6)
    assert np.allclose(g_rec[..., 0:3], r_rec[..., 0:3], rtol=1e-5, atol=1e-5)
    assert np.allclose(g_rec[..., 3], r_rec[..., 3], rtol=1e-5)
    if gate == 0.0:
        assert np.allclose(nis.cpu().numpy().reshape(r_nis.shape), r_nis, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('grid', [(60, 80), (68, 120)])
def test_kalman_scan_is_reproducible_under_memory_load(grid):
    """The production instantiations (no optional outputs; 60x80: double-buffered state, 68x120: single-buffer form whose
    records leave through 16-byte buffer stores with scalar descriptors) beside a loaded memory system: 20 launches, records
    and final state bit-identical to an unloaded launch and to the instantiation WITH the optional outputs.  (The first
    scalar-descriptor build of round 5 returned wrong records here: the store hazard of kfn_common.h buffer_store_b128.)"""
    import torch
    from tests.gpu_util import dev, stream
    from kfnet_amd import _lib
    lib = _lib.load()
    H, W = grid
    S, T = 48, 6
    rng = np.random.default_rng(H)
    flow = dev((rng.normal(size=(S, T, H, W, 2)) * 2.0).astype(np.float32))
    sig = dev(np.abs(rng.normal(size=(S, T, H, W, 1)) * 0.05).astype(np.float32))
    m = rng.normal(size=(S, T, H, W, 4)).astype(np.float32)
    m[..., 3] = np.abs(m[..., 3]) * 0.3 + 0.05
    meas = dev(m)
    st0 = rng.normal(size=(S, H, W, 4)).astype(np.float32)
    st0[..., 3] = np.abs(st0[..., 3]) * 0.3 + 0.05
    state0 = dev(st0)
    d = _lib.KalmanDesc(S=S, T=T, H=H, W=W, t0=1, reset_period=500, min_uncertainty=1e-5, nis_gate=0.0, has_transform=0)
    side = torch.cuda.Stream()
    big_a = torch.randn(64 << 20, device='cuda')
    big_b = torch.empty_like(big_a)

    def launch(load, debug=False):
        st = state0.clone()
        rec = torch.zeros(S * T * H * W * 4, device='cuda')
        tmp = torch.zeros(S * T * H * W * 4, device='cuda') if debug else None
        torch.cuda.synchronize()
        if load:
            _side_stream_load(torch, side, big_a, big_b, copies=2)
        _lib.check(lib.kfn_kalman_scan(C.byref(d), flow.data_ptr(), sig.data_ptr(), meas.data_ptr(), st.data_ptr(), rec.data_ptr(),
                                       tmp.data_ptr() if debug else None, None, None, stream()), 'scan')
        torch.cuda.synchronize()
        return rec, st

    ref_rec, ref_st = launch(False)
    dbg_rec, dbg_st = launch(False, debug=True)
    assert torch.equal(dbg_rec, ref_rec) and torch.equal(dbg_st, ref_st)
    bad = []
    for r in range(20):
        rec, st = launch(True)
        if not (torch.equal(rec, ref_rec) and torch.equal(st, ref_st)):
            bad.append(r)
    assert not bad, 'launches %s differ from the unloaded one' % bad


def test_kalman_lean_arithmetic_is_correctly_rounded():
    """The scan's square root and quotient (csrc/kfn_kalman.hip: sqrt_rn_normal / div_rn_normal -- the IEEE refinement steps without
    the denormal pre-scaling, because the scan is VALU-bound) against sqrtf and '/' of the same build at -ffp-contract=off, through
    kfn_kalman_arith_probe: BIT-IDENTICAL on 2^24 random bit patterns with exponents in [-90, 90] (both signs for the quotient), on
    this model's own range ([1e-10, 1e4]) and on the specials 0, inf and NaN; numpy's correctly rounded sqrt / divide agree too."""
    import torch
    from kfnet_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(2024)
    n = 1 << 24
    def rand_normal(size, signed):
        e = rng.integers(127 - 90, 127 + 90, size=size, dtype=np.uint32)
        m = rng.integers(0, 1 << 23, size=size, dtype=np.uint32)
        sgn = rng.integers(0, 2, size=size, dtype=np.uint32) if signed else np.zeros(size, np.uint32)
        return ((sgn << 31) | (e << 23) | m).view(np.float32)
    a = rand_normal(n, False)
    b = rand_normal(n, True)
    # exponents of a / b kept inside the normal range (the lean quotient's contract)
    keep = np.abs(np.log2(a.astype(np.float64)) - np.log2(np.abs(b).astype(np.float64))) < 100
    a, b = a[keep], b[keep]
    model = rng.uniform(1e-10, 1e4, size=1 << 20).astype(np.float32)
    a = np.concatenate([a, model, model[::-1].copy(), np.float32([0.0, 0.0, np.inf, 1.0, np.nan, 1.0, 4.0, 0.0])])
    b = np.concatenate([b, model[::-1].copy(), np.float32(1.0) + model, np.float32([1.0, 0.0, 2.0, np.inf, 1.0, np.nan, 0.0, -3.0])])
    n = a.size
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.zeros(n * 4, device='cuda')
    _lib.check(lib.kfn_kalman_arith_probe(da.data_ptr(), db.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream), 'probe')
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(n, 4)
    bits = o.view(np.uint32)
    nan_s = np.isnan(o[:, 0]) & np.isnan(o[:, 1])
    nan_q = np.isnan(o[:, 2]) & np.isnan(o[:, 3])
    assert np.all((bits[:, 0] == bits[:, 1]) | nan_s), 'lean sqrt differs from sqrtf in %d places' % int(np.sum((bits[:, 0] != bits[:, 1]) & ~nan_s))
    assert np.all((bits[:, 2] == bits[:, 3]) | nan_q), 'lean quotient differs from / in %d places' % int(np.sum((bits[:, 2] != bits[:, 3]) & ~nan_q))
    with np.errstate(all='ignore'):
        ref_s, ref_q = np.sqrt(a), a / b
    ok = np.isfinite(ref_q) & (np.abs(ref_q) >= np.finfo(np.float32).tiny) | (ref_q == 0) | np.isinf(ref_q)
    assert np.array_equal(o[~np.isnan(ref_s), 0], ref_s[~np.isnan(ref_s)])
    assert np.array_equal(o[ok, 2], ref_q[ok])


def test_kalman_fuse2_symmetric_variance():
    """kfn_kalman_fuse2 = KFNet.GetKFCoord2's fusion (KFNet/KFNet.py:487-502): bit exact vs the fp32 numpy oracle;
    KAT: equal noise -> K = 1/2, mean of the two, sigma / sqrt(2); its sigma equals BuildKFCoord's up to round-off."""
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(18)
    P = 4099
    pred = rng.normal(size=(P, 4)).astype(np.float32); pred[:, 3] = np.abs(pred[:, 3]) + 0.01
    meas = rng.normal(size=(P, 4)).astype(np.float32); meas[:, 3] = np.abs(meas[:, 3]) + 0.01
    meas[0, 3] = pred[0, 3]
    out = torch.zeros(P * 4, device='cuda'); out1 = torch.zeros(P * 4, device='cuda')
    dp, dm = dev(pred), dev(meas)
    _lib.check(lib.kfn_kalman_fuse2(dp.data_ptr(), dm.data_ptr(), out.data_ptr(), P, stream()), 'fuse2')
    _lib.check(lib.kfn_kalman_fuse(dp.data_ptr(), dm.data_ptr(), out1.data_ptr(), None, P, stream()), 'fuse')
    sync()
    o, o1 = out.cpu().numpy().reshape(P, 4), out1.cpu().numpy().reshape(P, 4)
    kx, ks = O.get_kf_coord2(pred[:, 0:3], pred[:, 3:4], meas[:, 0:3], meas[:, 3:4])
    assert np.array_equal(o[:, 0:3], kx) and np.array_equal(o[:, 3:4], ks)
    assert np.array_equal(o[:, 0:3], o1[:, 0:3])                      # same mean as BuildKFCoord
    assert np.isclose(o[0, 3], pred[0, 3] / np.sqrt(2), rtol=1e-6)
    ok = meas[:, 3] > 0.1 * pred[:, 3]            # (1-K)^2 P + K^2 R == (1-K) P in exact arithmetic; BuildKFCoord's form
    assert np.allclose(o[ok, 3], o1[ok, 3], rtol=1e-4)   # cancels when sigma_z << sigma^-, so compare where 1-K is not tiny
    assert lib.kfn_kalman_fuse2(dp.data_ptr() + 4, dm.data_ptr(), out.data_ptr(), P, stream()) == -1


def test_kalman_fuse_kat():
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(8)
    P = 5000
    pred = rng.normal(size=(P, 4)).astype(np.float32); pred[:, 3] = np.abs(pred[:, 3]) + 0.01
    meas = rng.normal(size=(P, 4)).astype(np.float32); meas[:, 3] = np.abs(meas[:, 3]) + 0.01
    meas[0, 3] = pred[0, 3]  # equal-noise KAT: K = 1/2
    out = torch.zeros(P * 4, device='cuda'); nis = torch.zeros(P * 3, device='cuda')
    dp, dm = dev(pred), dev(meas)
    _lib.check(lib.kfn_kalman_fuse(dp.data_ptr(), dm.data_ptr(), out.data_ptr(), nis.data_ptr(), P, stream()), 'fuse')
    sync()
    o = out.cpu().numpy().reshape(P, 4)
    kx, ks = O.build_kf_coord(pred[:, 0:3], pred[:, 3:4], meas[:, 0:3], meas[:, 3:4])
    assert np.array_equal(o[:, 0:3], kx) and np.array_equal(o[:, 3:4], ks)   # bit exact vs fp32 numpy
    assert np.allclose(o[0, 0:3], (pred[0, 0:3] + meas[0, 0:3]) / 2, rtol=1e-6)
    assert np.isclose(o[0, 3], pred[0, 3] / np.sqrt(2), rtol=1e-6)
    # property: min(s)/sqrt2 <= s_kf <= min(s).  The reference forms P = max(1-K,0)*P^- with
    # K rounded to fp32, so 1-K cancels catastrophically when sigma_z << sigma^-; the property
    # only holds to ~eps/(1-K) and is checked where 1-K is not tiny.
    mn = np.minimum(pred[:, 3], meas[:, 3])
    ok = (meas[:, 3] > 0.1 * pred[:, 3])
    assert np.all(o[ok, 3] <= mn[ok] * (1 + 1e-4)) and np.all(o[ok, 3] >= mn[ok] / np.sqrt(2) * (1 - 1e-4))
    rn = O.get_nis(meas[:, 0:3], meas[:, 3:4], pred[:, 0:3], pred[:, 3:4])
    assert np.allclose(nis.cpu().numpy().reshape(P, 3), rn, rtol=1e-6)


def test_conv_beyond_2gib_activations():
    """Activations larger than 2 GiB (32-bit buffer offsets are re-based per tile): every
    image of a batch of identical images must produce the bit-identical output."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_conv_kernel
    lib = _lib.load()
    N, H, W, ci, co = 30, 480, 640, 64, 32
    g = torch.Generator(device='cpu').manual_seed(3)
    img = torch.randn(H * W * ci, generator=g).cuda()
    x = img.repeat(N)                               # 2.36 GB
    assert x.numel() * 4 > 2 ** 31
    w = (np.random.default_rng(0).normal(size=(3, 3, ci, co)) / 24).astype(np.float32)
    wp = dev(pack_conv_kernel(w))
    y = torch.empty(N * H * W * co, device='cuda')
    d = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=32, ldy=co, kh=3, kw=3, stride=1, relu=1)
    _lib.check(lib.kfn_conv2d_nhwc(C.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), stream()), 'conv')
    sync()
    y = y.view(N, -1)
    assert float(y[0].abs().max()) > 0
    assert bool((y == y[0:1]).all())


@pytest.mark.parametrize('Cc', [16, 32, 4])
def test_flow_head_vs_oracle(Cc):
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(12)
    P = 517
    x = rng.normal(size=(P, 8, 8, Cc)).astype(np.float32)
    w = (rng.normal(size=(3, 3, Cc, 1)) * 0.8).astype(np.float32)
    b = np.array([0.3], np.float32)
    flow = torch.zeros(P * 2, device='cuda')
    logits = torch.zeros(P * 64, device='cuda')
    dx, dw, db = dev(x), dev(w[..., 0]), dev(b)
    _lib.check(lib.kfn_flow_head(dx.data_ptr(), dw.data_ptr(), db.data_ptr(), flow.data_ptr(), logits.data_ptr(),
                                 P, Cc, stream()), 'flow_head')
    sync()
    ref_logits = O.conv2d_same(x.astype(np.float64), w, b, 1, False)[..., 0].reshape(P, 64)
    pr = O.softmax(ref_logits)
    offs = O.coord_volume(np.zeros((1, 2, 2, 1)), np.zeros((1, 2, 2, 1)), 8)[1]
    # fp32 accumulation of 9*C terms: round-off scales with the logit magnitude
    tol = 3e-6 * max(1.0, float(np.abs(ref_logits).max()))
    assert np.abs(logits.cpu().numpy().reshape(P, 64) - ref_logits).max() < tol
    assert np.abs(flow.cpu().numpy().reshape(P, 2) - pr.dot(offs)).max() < 20 * tol


@pytest.mark.parametrize('P', [1, 5, 1029, 3001])
def test_oflow_tail_vs_oracle(P):
    """kfn_oflow_tail (conv6 + prediction + softmax + soft-argmax, one wave per window, patch resident in LDS)
    == the oracle's conv2d(relu) -> conv2d -> softmax -> offsets; window counts below / not a multiple of / far
    above the number of resident waves."""
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_oflow_tail_kernel
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(31 + P)
    x = np.maximum(rng.normal(size=(P, 8, 8, 48)), 0).astype(np.float32)
    w6 = (rng.normal(size=(3, 3, 48, 16)) * np.sqrt(2.0 / (9 * 48))).astype(np.float32)
    b6 = (rng.normal(size=16) * 0.1).astype(np.float32)
    wp = (rng.normal(size=(3, 3, 16, 1)) * 0.5).astype(np.float32)
    bp = np.array([0.3], np.float32)
    flow = torch.zeros(P * 2, device='cuda')
    logits = torch.zeros(P * 64, device='cuda')
    dx, d6, db6, dwp, dbp = dev(x), dev(pack_oflow_tail_kernel(w6)), dev(b6), dev(wp[..., 0]), dev(bp)
    _lib.check(lib.kfn_oflow_tail(dx.data_ptr(), d6.data_ptr(), db6.data_ptr(), dwp.data_ptr(), dbp.data_ptr(),
                                  flow.data_ptr(), logits.data_ptr(), P, 48, 16, stream()), 'oflow_tail')
    sync()
    mid = O.conv2d_same(x.astype(np.float64), w6, b6, 1, True)
    ref_logits = O.conv2d_same(mid, wp, bp, 1, False)[..., 0].reshape(P, 64)
    pr = O.softmax(ref_logits)
    offs = O.coord_volume(np.zeros((1, 2, 2, 1)), np.zeros((1, 2, 2, 1)), 8)[1]
    tol = 6e-6 * max(1.0, float(np.abs(ref_logits).max()))
    assert np.abs(logits.cpu().numpy().reshape(P, 64) - ref_logits).max() < tol
    assert np.abs(flow.cpu().numpy().reshape(P, 2) - pr.dot(offs)).max() < 20 * tol
    # without the optional logits output, and a shape the kernel declines
    flow2 = torch.zeros(P * 2, device='cuda')
    _lib.check(lib.kfn_oflow_tail(dx.data_ptr(), d6.data_ptr(), db6.data_ptr(), dwp.data_ptr(), dbp.data_ptr(),
                                  flow2.data_ptr(), None, P, 48, 16, stream()), 'oflow_tail')
    sync()
    assert torch.equal(flow, flow2)
    assert lib.kfn_oflow_tail(dx.data_ptr(), d6.data_ptr(), None, dwp.data_ptr(), None, flow2.data_ptr(), None, P, 32, 16,
                              stream()) == -3


def test_window_fc_conv_on_2x2_windows():
    """WindowFcConvOp's launch: a 3x3 SAME conv on [P,2,2,Cin] windows as the 1x1 convolution of [P,1,1,4 Cin] with
    the dense window matrix (pack_window_fc_kernel) == the oracle's nine-tap convolution."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_bias_x4, pack_window_fc_kernel
    lib = _lib.load()
    rng = np.random.default_rng(44)
    P, ci, co = 1037, 128, 64
    x = np.maximum(rng.normal(size=(P, 2, 2, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    d = _lib.ConvDesc(N=P, H=1, W=1, Cin=4 * ci, ldx=4 * ci, Cout=4 * co, cout_pad=4 * co, ldy=4 * co, kh=1, kw=1,
                      stride=1, relu=1)
    y = torch.zeros((P, 4 * co), device='cuda')
    dx, dw, db = dev(x), dev(pack_window_fc_kernel(wt)), dev(pack_bias_x4(b))
    _lib.check(lib.kfn_conv2d_nhwc(C.byref(d), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), y.data_ptr(), stream()), 'fc')
    sync()
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 1, True)
    _check_err(np.abs(y.cpu().numpy().reshape(ref.shape) - ref).max(), x, wt, 'direct', 'window fc 2x2')


WINO_CASES = [(1, 8, 8, 128, 128), (2, 7, 9, 128, 160), (1, 60, 80, 256, 128), (3, 12, 16, 64, 36),
              (1, 5, 5, 32, 4), (2, 10, 6, 48, 64)]


@pytest.mark.parametrize('case', WINO_CASES)
@pytest.mark.parametrize('config', [0, 1, 3, 6, 9])
def test_winograd_conv_vs_oracle(case, config):
    """kfn_conv2d_winograd == kfn_conv2d_nhwc == oracle up to fp32 round-off (incl. odd sizes,
    where the last 2x2 tile overhangs, strided output, bias + ReLU)."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_kernel
    lib = _lib.load()
    n, h, w, ci, co = case
    rng = np.random.default_rng(n * 1000 + h * 10 + ci)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ldy = co + 8
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=ldy, kh=3, kw=3,
                      stride=1, relu=1, config=config)
    nb = C.c_size_t()
    _lib.check(lib.kfn_winograd_workspace_bytes(C.byref(d), C.byref(nb)), 'ws')
    GUARD = 4096   # floats behind the workspace / rows behind the output: partial tiles must not spill
    ws = torch.full((nb.value // 4 + GUARD,), -7.0, device='cuda')
    y = torch.full((n * h * w + 64, ldy), -5.0, device='cuda')
    dx, du, db = dev(x), dev(pack_winograd_kernel(wt)), dev(b)
    _lib.check(lib.kfn_conv2d_winograd(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(),
                                       ws.data_ptr(), 3, stream()), 'wino')
    sync()
    assert bool((ws[nb.value // 4:] == -7.0).all()), 'Winograd GEMMs wrote past the workspace'
    got = y.cpu().numpy()
    assert np.all(got[n * h * w:] == -5.0), 'output transform wrote past the last pixel'
    got = got[:n * h * w]
    assert np.all(got[:, co:] == -5.0)
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 1, True)
    err = np.abs(got[:, :co].reshape(ref.shape) - ref).max()
    _check_err(err, x, wt, 'f23', 'wino two-kernel %s cfg %d' % (case, config))


@pytest.mark.parametrize('shape', [(2, 7, 9, 32, 32), (1, 60, 80, 32, 32), (3, 5, 4, 16, 48)])
def test_cost_volume_conv_vs_oracle(shape):
    """kfn_cost_volume_conv == conv0(BuildCoordVolume(f1, f2)) of the oracle (volume generated
    in the loader, incl. image-border zero fill of the shifted f1 and conv0's own SAME padding
    at the window border), written into a channel window of a wider buffer."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_conv_kernel
    lib = _lib.load()
    N, H, W, Cc, co = shape
    rng = np.random.default_rng(77)
    f = rng.normal(size=(N + 1, H, W, Cc)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, Cc, co)) / np.sqrt(9 * Cc)).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ldy, off = co + 16, 16
    fd, wd, bd = dev(f), dev(pack_conv_kernel(wt)), dev(b)
    y = torch.full((N * H * W * 64, ldy), -9.0, device='cuda')
    _lib.check(lib.kfn_cost_volume_conv(fd.data_ptr(), fd.data_ptr() + H * W * Cc * 4, wd.data_ptr(), bd.data_ptr(),
                                        y.data_ptr() + off * 4, N, H, W, Cc, co, -(-co // 32) * 32, ldy, 1, 0,
                                        stream()), 'cvconv')
    sync()
    got = y.cpu().numpy()
    assert np.all(got[:, :off] == -9.0)
    for n in range(N):
        vol, _ = O.coord_volume(f[n:n + 1].astype(np.float64), f[n + 1:n + 2].astype(np.float64), 8)
        ref = O.conv2d_same(vol, wt, b, 1, True).reshape(H * W * 64, co)
        g = got[n * H * W * 64:(n + 1) * H * W * 64, off:off + co]
        assert np.abs(g - ref).max() < 2e-5


F16_CASES = [(1, 8, 8, 32, 32, 3, 1), (2, 13, 17, 64, 48, 3, 1), (1, 12, 20, 64, 64, 3, 2), (1, 15, 9, 32, 100, 3, 2),
             (2, 6, 10, 256, 128, 1, 1), (1, 60, 80, 128, 160, 3, 1), (6, 2, 2, 64, 64, 3, 1)]


@pytest.mark.parametrize('case', F16_CASES)
@pytest.mark.parametrize('config', [0, 1, 2, 4, 6])
def test_conv_fp16_operands(case, config):
    """operand_dtype = F16: the kernel must equal an fp64 convolution of the fp16-ROUNDED
    operands (products of halfs are exact in fp32, accumulation is fp32), i.e. the only error
    vs that reference is fp32 summation order; vs the unrounded fp32 conv it is ~1e-3 relative."""
    from tests.gpu_util import run_conv
    n, h, w, ci, co, k, s = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.normal(size=(n, h, w, ci)).astype(np.float32)
    wt = (rng.normal(size=(k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    y = run_conv(x, wt, b, s, True, config=config, f16=True)
    xr = x.astype(np.float16).astype(np.float64)
    wr = wt.astype(np.float16).astype(np.float64)
    ref = O.conv2d_same(xr, wr, b, s, True)
    _check_err(np.abs(y - ref).max(), x, wt, 'direct', 'conv f16 operands %s cfg %d' % (case, config))
    full = O.conv2d_same(x.astype(np.float64), wt, b, s, True)
    assert np.abs(y - full).max() < 2e-2 and np.abs(y - full).max() > 0   # it really is reduced precision


F16_ACT_CASES = [
    # N, H, W, Cin, Cout, k, stride
    (1, 16, 24, 64, 64, 3, 1),        # conv1b class: Cout 64 -> the 192x64 tile, one 64-channel stage per tap
    (2, 13, 17, 64, 256, 3, 2),       # conv2a class: stride 2, odd size (pad (1,1)), partial last row tile
    (1, 12, 20, 256, 256, 3, 1),      # conv2b class: two column tiles
    (1, 9, 11, 128, 72, 3, 1),        # Cout % 64 != 0: the chunks past Cout of the last column tile are dropped
    (3, 6, 10, 256, 128, 1, 1),       # conv7 class: 1x1
    (1, 60, 80, 96, 160, 3, 1),       # Cin % 64 != 0 -> k-step 16; M = 4800
    (7, 1, 1, 128, 128, 3, 1),        # 1x1 images: only the centre tap is live
]


@pytest.mark.parametrize('case', F16_ACT_CASES)
@pytest.mark.parametrize('config,k_step', [(0, 0), (2, 16), (2, 32), (9, 16), (9, 32), (7, 0), (3, 0), (12, 16), (12, 32)])
def test_conv_fp16_activations(case, config, k_step):
    """x_dtype = y_dtype = KFN_ACT_F16 (BASELINE config 5, fp16 activations end to end): the input tensor holds
    halfs, the output is ONE RNE rounding of the fp32 result.  Reference: an fp64 convolution of the fp16 input and
    fp16-rounded weights; tolerance = the fp32 accumulation bound + half an fp16 ulp of the result."""
    from tests.gpu_util import run_conv
    n, h, w, ci, co, k, s = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.normal(size=(n, h, w, ci)).astype(np.float16)
    wt = (rng.normal(size=(k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    y = run_conv(x, wt, b, s, True, config=config, f16=True, x16=True, y16=True, k_step=k_step)
    ref = O.conv2d_same(x.astype(np.float64), wt.astype(np.float16).astype(np.float64), b, s, True)
    tol = _conv_tol(x.astype(np.float32), wt) + np.abs(ref) * 2.0 ** -11 + 1e-7
    assert y.shape == ref.shape
    assert np.all(np.abs(y - ref) <= tol), float((np.abs(y - ref) - tol).max())


C64_CASES = [
    # N, H, W, ldx, ldy, x_off, y_off, relu, bias
    (1, 16, 24, 64, 64, 0, 0, True, True),       # one 128-pixel strip, mostly padding
    (2, 13, 200, 64, 64, 0, 0, True, True),      # two 128-pixel strips, ragged second one; odd height
    (1, 9, 192, 64, 64, 0, 0, False, True),      # one full 192-pixel strip, no ReLU
    (3, 5, 131, 72, 80, 8, 16, True, True),      # strided views in and out (ldx / ldy in elements), 3 images
    (1, 1, 70, 64, 64, 0, 0, True, False),       # a single row: both vertical taps read zero rows; no bias
    (2, 2, 385, 64, 64, 0, 0, True, True),       # 385 = 2 x 192 + 1: a strip with one live column
    (1, 37, 960, 64, 64, 0, 0, True, True),      # BASELINE config 5's width: 5 strips of 192, several chunks of rows
    (1, 67, 320, 64, 64, 0, 0, True, True),      # many chunks: the chunk seams (halo rows recomputed by the neighbour)
]


@pytest.mark.parametrize('case', C64_CASES)
def test_conv3x3_c64_f16_vs_oracle(case):
    """kfn_conv3x3_c64_f16 (csrc/kfn_conv64.hip: weights in registers, a workgroup walks down a pixel strip, every input row
    feeds three output rows' accumulators) == an fp64 convolution of the fp16 input with the fp16-rounded weights, to the fp32
    accumulation bound + half an fp16 ulp (the output is ONE RNE rounding of the fp32 result) -- the bar of
    test_conv_fp16_activations; guard rows and the untouched channel columns keep their sentinels."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_conv64_rows_kernel
    lib = _lib.load()
    n, h, w, ldx, ldy, x_off, y_off, relu, with_bias = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.normal(size=(n, h, w, 64)).astype(np.float16)
    wt = (rng.normal(size=(3, 3, 64, 64)) / np.sqrt(9 * 64)).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32) if with_bias else None
    xb = np.full((n * h * w, ldx), 7.0, dtype=np.float16)
    xb[:, x_off:x_off + 64] = x.reshape(-1, 64)
    xd = dev(xb)
    GUARD = 8
    yd = torch.full((n * h * w + GUARD, ldy), -123.0, dtype=torch.float16, device='cuda')
    wd_, bd = dev(pack_conv64_rows_kernel(wt)), (dev(b) if with_bias else None)
    assert wd_.dtype == torch.float16 and tuple(wd_.shape) == (2, 36, 64, 8)
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=64, ldx=ldx, Cout=64, cout_pad=64, ldy=ldy, kh=3, kw=3, stride=1, relu=int(relu),
                      operand_dtype=_lib.OPERAND_F16, x_dtype=_lib.ACT_F16, y_dtype=_lib.ACT_F16)
    assert lib.kfn_conv3x3_c64_f16_supported(C.byref(d)) == 1
    for _ in range(2):      # twice: a race between the row ring, the output tiles and the barrier would not repeat
        _lib.check(lib.kfn_conv3x3_c64_f16(C.byref(d), xd.data_ptr() + 2 * x_off, wd_.data_ptr(),
                                           bd.data_ptr() if bd is not None else None, yd.data_ptr() + 2 * y_off, stream()), 'c64')
        sync()
        yh = yd.cpu().numpy().astype(np.float32)
        assert np.all(yh[n * h * w:] == -123.0)
        mask = np.ones(ldy, dtype=bool)
        mask[y_off:y_off + 64] = False
        assert np.all(yh[:n * h * w][:, mask] == -123.0)
        got = yh[:n * h * w, y_off:y_off + 64].reshape(n, h, w, 64)
        ref = O.conv2d_same(x.astype(np.float64), wt.astype(np.float16).astype(np.float64), b, 1, relu)
        tol = _conv_tol(x.astype(np.float32), wt) + np.abs(ref) * 2.0 ** -11 + 1e-7
        assert np.all(np.abs(got - ref) <= tol), float((np.abs(got - ref) - tol).max())
        yd.fill_(-123.0)


def _side_stream_load(torch, side, big_a, big_b, copies=6):
    """512 MB copies on another stream: HBM / fabric back-pressure beside the kernel under test."""
    with torch.cuda.stream(side):
        for _ in range(copies):
            big_b.copy_(big_a)


def test_conv3x3_c64_f16_is_reproducible_under_memory_load():
    """Regression test of the 16-byte buffer-store hazard (kfn_common.h buffer_store_b128): round 4's kernel sent,
    for one launch in seven of this shape while another stream loaded the memory system, the NEXT store's byte offset as the
    first dword of a few 16-byte pieces (the compiler reuses a store's dead data register at once; the store reads it late).
    40 launches beside 3 GB of copies each: all bit-identical to the first AND to an unloaded launch."""
    import torch
    from tests.gpu_util import stream
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_conv64_rows_kernel
    lib = _lib.load()
    n, h, w = 5, 540, 960
    rng = np.random.default_rng(3)
    x = torch.from_numpy(np.maximum(rng.normal(size=(n, h, w, 64)), 0).astype(np.float16)).cuda()
    wt = (rng.normal(size=(3, 3, 64, 64)) * np.sqrt(2.0 / 576)).astype(np.float32)
    wp = torch.from_numpy(pack_conv64_rows_kernel(wt)).cuda()
    b = torch.from_numpy(rng.normal(size=64).astype(np.float32)).cuda()
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=64, ldx=64, Cout=64, cout_pad=64, ldy=64, kh=3, kw=3, stride=1, relu=1,
                      operand_dtype=_lib.OPERAND_F16, x_dtype=_lib.ACT_F16, y_dtype=_lib.ACT_F16)
    side = torch.cuda.Stream()
    big_a = torch.randn(64 << 20, device='cuda')
    big_b = torch.empty_like(big_a)

    def launch(load):
        y = torch.full((n * h * w, 64), -3.0, dtype=torch.float16, device='cuda')
        torch.cuda.synchronize()
        if load:
            _side_stream_load(torch, side, big_a, big_b)
        _lib.check(lib.kfn_conv3x3_c64_f16(C.byref(d), x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), stream()), 'c64')
        torch.cuda.synchronize()
        return y

    ref = launch(False)
    bad = [r for r in range(40) if not torch.equal(launch(True), ref)]
    assert not bad, 'launches %s differ from the unloaded one' % bad


@pytest.mark.parametrize('kind,form', [('f43', 3), ('s2', 4), ('s2', 5), ('fused', 0)])
def test_winograd_16_byte_stores_are_reproducible_under_memory_load(kind, form):
    """The same condition for the Winograd epilogues (their 16-byte image-row stores carry an SGPR row offset too)."""
    import torch
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_f43_kernel_b, pack_winograd_fused_kernel, pack_winograd_s2_kernel_b, pack_winograd_s2_kernel_c
    from tests.gpu_util import stream
    lib = _lib.load()
    n, h, w, ci, co = 8, 120, 160, 128, 128
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.normal(size=(n, h, w, ci)).astype(np.float32)).cuda()
    wt = (rng.normal(size=(3, 3, ci, co)) / np.sqrt(9 * ci)).astype(np.float32)
    stride = 2 if kind == 's2' else 1
    pack = {'f43': pack_winograd_f43_kernel_b, 's2': pack_winograd_s2_kernel_c if form == 5 else pack_winograd_s2_kernel_b,
            'fused': pack_winograd_fused_kernel}[kind]
    entry = {'f43': lib.kfn_conv2d_winograd_f43, 's2': lib.kfn_conv2d_winograd_s2, 'fused': lib.kfn_conv2d_winograd_fused}[kind]
    u = torch.from_numpy(pack(wt)).cuda()
    b = torch.from_numpy(rng.normal(size=co).astype(np.float32)).cuda()
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=3, kw=3, stride=stride, relu=1, wino_form=form)
    side = torch.cuda.Stream()
    big_a = torch.randn(64 << 20, device='cuda')
    big_b = torch.empty_like(big_a)

    def launch(load):
        y = torch.full((n * (h // stride) * (w // stride), co), -3.0, device='cuda')
        torch.cuda.synchronize()
        if load:
            _side_stream_load(torch, side, big_a, big_b, copies=3)
        _lib.check(entry(C.byref(d), x.data_ptr(), u.data_ptr(), b.data_ptr(), y.data_ptr(), stream()), kind)
        torch.cuda.synchronize()
        return y

    ref = launch(False)
    bad = [r for r in range(25) if not torch.equal(launch(True), ref)]
    assert not bad, '%s: launches %s differ from the unloaded one' % (kind, bad)


def test_conv3x3_c64_f16_rejects_what_it_cannot_do():
    from kfnet_amd import _lib
    lib = _lib.load()
    ok = dict(N=1, H=8, W=8, Cin=64, ldx=64, Cout=64, cout_pad=64, ldy=64, kh=3, kw=3, stride=1, relu=1,
              operand_dtype=_lib.OPERAND_F16, x_dtype=_lib.ACT_F16, y_dtype=_lib.ACT_F16)
    assert lib.kfn_conv3x3_c64_f16_supported(C.byref(_lib.ConvDesc(**ok))) == 1
    for bad in (dict(Cin=32, ldx=32), dict(Cout=128, cout_pad=128, ldy=128), dict(stride=2), dict(kh=1, kw=1), dict(x_dtype=0),
                dict(y_dtype=0), dict(operand_dtype=0), dict(ldx=68), dict(ldy=66), dict(transposed=1), dict(epilogue=1)):
        d = _lib.ConvDesc(**dict(ok, **bad))
        assert lib.kfn_conv3x3_c64_f16_supported(C.byref(d)) == 0, bad
        assert lib.kfn_conv3x3_c64_f16(C.byref(d), 16, 16, None, 16, None) == -3, bad      # KFN_ERR_UNSUPPORTED, before any launch


@pytest.mark.parametrize('case', F16_ACT_CASES + [(2, 30, 40, 512, 512, 3, 1), (1, 17, 23, 256, 320, 3, 2),
                                                  (2, 68, 120, 1024, 512, 3, 1)])
@pytest.mark.parametrize('config,path', [(2, 2), (9, 2), (2, 3), (9, 3), (12, 3)])
def test_conv_fp16_activations_weights_via_lds_dma(case, config, path):
    """weights_path = KFN_WEIGHTS_LDS_DMA (2: the weight tile goes global -> LDS directly, swizzle applied at the source,
    three weight buffers) and KFN_OPERANDS_LDS_DMA (3: the activation tile too, zero padding = out-of-range lanes
    writing zeros, explicit vmcnt(0) in front of every barrier): BIT-identical to the register-staged path -- same operands, same MFMA order -- on shapes
    with 1 to 144 K stages, partial row / column tiles and Cout below the tile width; run twice (a race between a
    transfer and the fragment reads or the epilogue overlay would not be deterministic)."""
    from tests.gpu_util import run_conv
    n, h, w, ci, co, k, s = case
    if ci % 32 != 0:
        pytest.skip('fp16 operands need Cin % 32 == 0')
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.normal(size=(n, h, w, ci)).astype(np.float16)
    wt = (rng.normal(size=(k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ref = run_conv(x, wt, b, s, True, config=config, f16=True, x16=True, y16=True, k_step=16, weights_path=1)
    for _ in range(3):
        y = run_conv(x, wt, b, s, True, config=config, f16=True, x16=True, y16=True, k_step=16, weights_path=path)
        assert np.array_equal(y, ref)


@pytest.mark.parametrize('x16,y16', [(True, False), (False, True)])
def test_conv_fp16_activations_mixed_and_strided(x16, y16):
    """fp16 in -> fp32 out (the 'prediction' head, with its exp epilogue on a narrow tile) and fp32 in -> fp16 out,
    through strided views (ldx / ldy count ELEMENTS, the untouched columns must keep their contents)."""
    from kfnet_amd import _lib
    from tests.gpu_util import run_conv
    rng = np.random.default_rng(77)
    ci, co = 128, (4 if x16 else 64)
    x = rng.normal(size=(2, 9, 12, ci)).astype(np.float16 if x16 else np.float32)
    wt = (rng.normal(size=(1, 1, ci, co)) / np.sqrt(ci)).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    y = run_conv(x, wt, b, 1, not x16, f16=True, x16=x16, y16=y16, ldx=ci + 64, x_off=32, ldy=co + 24, y_off=8,
                 epilogue=_lib.EPI_EXP_CH3 if x16 else 0)
    xr = x.astype(np.float16).astype(np.float64)
    ref = O.conv2d_same(xr, wt.astype(np.float16).astype(np.float64), b, 1, not x16)
    if x16:
        ref[..., 3] = np.exp(ref[..., 3])
    tol = _conv_tol(x.astype(np.float32), wt) * (4.0 if x16 else 1.0) + (np.abs(ref) * 2.0 ** -11 if y16 else 0.0)
    assert np.all(np.abs(y - ref) <= tol)


@pytest.mark.parametrize('W', [70, 132])
def test_first_conv_fp16_head(W):
    """kfn_first_conv_u8_ex with an fp16 first head (config 5: SCoordNet conv1a) beside an fp32 second head: the
    fp16 head equals the fp32 kernel's result rounded once; the second head is bit-identical to the fp32 call.
    W = 70: byte-wise kernel, W = 132: lane-per-pixel kernel."""
    import torch
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_first_kernel
    from tests.gpu_util import dev, stream, sync
    lib = _lib.load()
    rng = np.random.default_rng(3)
    N, H = 2, 37
    img = rng.integers(0, 256, size=(N, H, W, 3)).astype(np.uint8)
    w1 = (rng.normal(size=(3, 3, 3, 64)) / 5).astype(np.float32)
    w2 = (rng.normal(size=(3, 3, 3, 16)) / 5).astype(np.float32)
    b1, b2 = rng.normal(size=64).astype(np.float32), rng.normal(size=16).astype(np.float32)
    di, dw1, dw2, db1, db2 = dev(img), dev(pack_first_kernel(w1)), dev(pack_first_kernel(w2)), dev(b1), dev(b2)
    y1 = torch.empty((N, H, W, 64), dtype=torch.float32, device='cuda')
    y2 = torch.empty((N, H, W, 16), dtype=torch.float32, device='cuda')
    _lib.check(lib.kfn_first_conv_u8(di.data_ptr(), N, H, W, dw1.data_ptr(), db1.data_ptr(), y1.data_ptr(), 64,
                                     dw2.data_ptr(), db2.data_ptr(), y2.data_ptr(), 16, stream()), 'first')
    h1 = torch.full((N, H, W, 64), -5.0, dtype=torch.float16, device='cuda')
    z2 = torch.empty((N, H, W, 16), dtype=torch.float32, device='cuda')
    _lib.check(lib.kfn_first_conv_u8_ex(di.data_ptr(), N, H, W, dw1.data_ptr(), db1.data_ptr(), h1.data_ptr(), 64,
                                        _lib.ACT_F16, dw2.data_ptr(), db2.data_ptr(), z2.data_ptr(), 16, stream()), 'first16')
    sync()
    assert torch.equal(z2, y2)
    assert torch.equal(h1, y1.half())
    ref = O.conv2d_same(O.preprocess(img, np.float64), w1, b1, 1, True)
    assert np.abs(y1.cpu().numpy() - ref).max() < 1e-4


def test_fp16_activation_descriptor_validation():
    """kfn_conv2d_nhwc tells the host what it cannot do with fp16 activations (KFN_ERR_ARG = -1) instead of computing
    something else: fp32 operands, transposed convs, misaligned strides / pointers, head epilogues on fp16 output."""
    import torch
    from kfnet_amd import _lib
    from tests.gpu_util import stream
    lib = _lib.load()
    x = torch.zeros(2 * 8 * 8 * 64, dtype=torch.float16, device='cuda')
    w = torch.zeros(64 * 9 * 64, dtype=torch.float16, device='cuda')
    y = torch.zeros(2 * 8 * 8 * 64 + 8, dtype=torch.float16, device='cuda')
    base = dict(N=2, H=8, W=8, Cin=64, ldx=64, Cout=64, cout_pad=64, ldy=64, kh=3, kw=3, stride=1, relu=1,
                operand_dtype=_lib.OPERAND_F16, x_dtype=_lib.ACT_F16, y_dtype=_lib.ACT_F16)
    def call(yoff=0, **kw):
        d = _lib.ConvDesc(**dict(base, **kw))
        return lib.kfn_conv2d_nhwc(C.byref(d), x.data_ptr(), w.data_ptr(), None, y.data_ptr() + yoff, stream())
    assert call() == 0
    assert call(operand_dtype=_lib.OPERAND_F32) == -1
    assert call(stride=2, transposed=1) == -1
    assert call(ldx=68) == -1 and call(ldy=68) == -1 and call(Cout=60, ldy=64) == -1
    assert call(yoff=2) == -1
    assert call(epilogue=_lib.EPI_EXP_CH3) == -1
    assert call(x_dtype=7) == -1 and call(k_step=24) == -1
    assert call(config=1) == -1            # 160x128 has no fp16-activation instantiation
    assert b'config' in lib.kfn_last_error()
    torch.cuda.synchronize()


def test_deconv_fp16_operands():
    from tests.gpu_util import run_conv
    rng = np.random.default_rng(31)
    x = rng.normal(size=(40, 4, 4, 32)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, 16, 32)) / 17).astype(np.float32)
    b = rng.normal(size=16).astype(np.float32)
    y = run_conv(x, wt, b, 2, True, transposed=True, f16=True)
    ref = O.conv2d_transpose_same(x.astype(np.float16).astype(np.float64), wt.astype(np.float16).astype(np.float64), b, 2, True)
    _check_err(np.abs(y - ref).max(), x, wt, 'direct', 'deconv f16 operands', transposed=True)


@pytest.mark.parametrize('case', F16_CASES)
@pytest.mark.parametrize('config', [0, 1, 2, 6])
def test_conv_f16x3_split_operands(case, config):
    """operand_dtype = F16X3 (hi/lo split, 3 fp16 MFMA products, fp32 accumulate) must be
    fp32-class accurate against the fp64 convolution of the ORIGINAL fp32 operands."""
    from tests.gpu_util import run_conv
    n, h, w, ci, co, k, s = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 1)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32) * 3
    wt = (rng.normal(size=(k, k, ci, co)) * np.sqrt(2.0 / (k * k * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    y = run_conv(x, wt, b, s, True, config=config, x3=True)
    ref = O.conv2d_same(x.astype(np.float64), wt, b, s, True)
    d32 = O.conv2d_same(x, wt, b, s, True)
    err, err32 = np.abs(y - ref).max(), np.abs(d32 - ref).max()
    print('f16x3 err %.3g vs plain fp32 err %.3g (scale %.3g)' % (err, err32, np.abs(ref).max()))
    _check_err(err, x, wt, 'f16x3', 'conv f16x3 %s cfg %d' % (case, config))


# (N, H, W, Cin, Cout): multiples of the 8x4 tile block and ragged ones, blocks that straddle two
# images of the batch (Th % 4 != 0), a single image shorter than the batch packing, odd sizes where
# the last 2x2 tile overhangs, Cout not a multiple of 32, wide K
FUSED_CASES = [(1, 8, 8, 128, 128), (2, 7, 9, 128, 160), (1, 60, 80, 256, 128), (3, 12, 16, 64, 36),
               (2, 10, 6, 48, 64), (2, 30, 40, 64, 96), (5, 60, 80, 32, 32), (3, 9, 70, 16, 32),
               (4, 14, 33, 32, 40), (1, 64, 96, 64, 64), (17, 10, 12, 16, 8), (2, 16, 30, 1024, 32),
               # >= 128 output channels and Cin % 32 == 0: the four-wave form (kfn_wino3.hip) -- ragged channel
               # tiles (waves past Cout), blocks straddling two images, ragged tile blocks, many super-steps
               (5, 30, 40, 64, 256), (3, 14, 33, 32, 192), (17, 10, 12, 32, 136), (2, 12, 16, 512, 128),
               # 33 .. 64 output channels and Cin % 16 == 0: the two-wave form of kfn_wino3.hip ((3,12,16,64,36),
               # (2,10,6,48,64), (4,14,33,32,40), (1,64,96,64,64) above take it too) -- one and five super-steps
               (17, 10, 12, 16, 64), (2, 30, 40, 80, 64)]


@pytest.mark.parametrize('relu', [1, 0])
@pytest.mark.parametrize('case', FUSED_CASES)
def test_winograd_fused_vs_oracle(case, relu):
    """kfn_conv2d_winograd_fused (one wavefront = 8x4 tiles x 32 channels x all 16 positions, both
    transforms in registers, no workspace) == oracle up to fp32 round-off; strided output window,
    guard rows behind the tensor untouched."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_fused_kernel
    lib = _lib.load()
    n, h, w, ci, co = case
    rng = np.random.default_rng(n * 1000 + h * 10 + ci + 5)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ldy = co + 8
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=ldy, kh=3, kw=3,
                      stride=1, relu=relu)
    assert lib.kfn_winograd_fused_supported(C.byref(d)) == 1
    GUARD = 64
    y = torch.full((n * h * w + GUARD, ldy), -5.0, device='cuda')
    dx, du, db = dev(x), dev(pack_winograd_fused_kernel(wt)), dev(b)
    _lib.check(lib.kfn_conv2d_winograd_fused(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(),
                                             stream()), 'wino2')
    sync()
    got = y.cpu().numpy()
    assert np.all(got[:, co:] == -5.0) and np.all(got[n * h * w:] == -5.0)
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 1, bool(relu))
    err = np.abs(got[:n * h * w, :co].reshape(ref.shape) - ref).max()
    _check_err(err, x, wt, 'f23', 'wino fused %s relu %d' % (case, relu))


# (N, H, W, Cin, Cout): whole blocks, blocks straddling two images (Th % 8 != 0), ragged tile columns / rows (H, W not
# multiples of 4 or 16), channel counts that leave waves past Cout (Cout % 64 != 0), one and many super-steps
F43_CASES = [(2, 32, 16, 16, 64), (3, 60, 80, 32, 64), (2, 30, 40, 64, 128), (5, 36, 20, 16, 96), (1, 68, 120, 48, 72),
             (3, 29, 35, 32, 100), (2, 120, 160, 16, 64), (9, 32, 16, 512, 64), (2, 60, 80, 1024, 128),
             # narrow images: fewer tile columns than a block holds (Tw = 2 / 3 < 4) -- ADVICE r4
             (2, 32, 8, 16, 64), (1, 29, 12, 32, 72)]


@pytest.mark.parametrize('form', [2, 3])      # KFN_WINO_FORM_F43_FOUR_WAVE (wino4_kernel), _EIGHT_WAVE (wino4b_kernel)
@pytest.mark.parametrize('relu', [1, 0])
@pytest.mark.parametrize('case', F43_CASES)
def test_winograd_f43_vs_oracle(case, relu, form):
    """kfn_conv2d_winograd_f43 (F(4x4,3x3): 36 positions over the waves of a workgroup -- four waves on 32x32x2 MFMA tiles or
    eight on 16x16x4 --, the xi half of the output transform reduced across waves through LDS) == oracle up to fp32
    round-off; strided input and output windows, guard rows behind the tensor untouched.  Tolerance: tests/conv_tol.py's
    model with the F(4x4,3x3) constant (4.0e-4 on O(1) outputs at Cin = 1024, where the kernels measure 1.9e-4)."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_f43_kernel, pack_winograd_f43_kernel_b
    lib = _lib.load()
    n, h, w, ci, co = case
    rng = np.random.default_rng(n * 1000 + h * 10 + ci + 43)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ldx, ldy = ci + 8, co + 8
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ldx, Cout=co, cout_pad=-(-co // 32) * 32, ldy=ldy, kh=3, kw=3,
                      stride=1, relu=relu, wino_form=form)
    assert lib.kfn_winograd_f43_supported(C.byref(d)) == 1
    GUARD = 64
    xb = np.full((n * h * w, ldx), 9.0, dtype=np.float32)      # the columns behind Cin must never be read into the sum
    xb[:, :ci] = x.reshape(-1, ci)
    y = torch.full((n * h * w + GUARD, ldy), -5.0, device='cuda')
    pack = pack_winograd_f43_kernel_b if form == 3 else pack_winograd_f43_kernel      # the eight-wave form reads pairs of positions
    dx, du, db = dev(xb), dev(pack(wt)), dev(np.concatenate([b, np.zeros((-co) % 4, np.float32)]))
    _lib.check(lib.kfn_conv2d_winograd_f43(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(),
                                           stream()), 'wino4')
    sync()
    got = y.cpu().numpy()
    assert np.all(got[:, co:] == -5.0) and np.all(got[n * h * w:] == -5.0)
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 1, bool(relu))
    err = np.abs(got[:n * h * w, :co].reshape(ref.shape) - ref).max()
    _check_err(err, x, wt, 'f43', 'wino f43 %s relu %d form %d' % (case, relu, form))


def test_winograd_f43_rejects_what_it_cannot_do():
    from kfnet_amd import _lib
    lib = _lib.load()
    ok = dict(N=1, H=32, W=32, Cin=32, ldx=32, Cout=64, cout_pad=64, ldy=64, kh=3, kw=3, stride=1)
    assert lib.kfn_winograd_f43_supported(C.byref(_lib.ConvDesc(**ok))) == 1
    for bad in (dict(H=28), dict(Cin=24, ldx=24), dict(stride=2), dict(Cout=62, ldy=62), dict(ldy=66), dict(x_dtype=1),
                dict(operand_dtype=1), dict(epilogue=2), dict(kh=1, kw=1),
                # ... and everything else the launcher would refuse without seeing the pointers
                dict(ldx=16), dict(ldy=60), dict(ldx=33), dict(cout_pad=32), dict(N=0), dict(H=8192, W=8192, Cin=16, ldx=16)):
        assert lib.kfn_winograd_f43_supported(C.byref(_lib.ConvDesc(**dict(ok, **bad)))) == 0, bad


@pytest.mark.parametrize('case', [(3, 14, 33, 64, 64, 0), (2, 30, 40, 48, 40, 8), (5, 30, 40, 64, 256, 0)])
def test_winograd_fused_forms_agree(case):
    """kfn_conv_desc.wino_form = KFN_WINO_FORM_ONE_WAVE (wino2_kernel) against the default route (two-wave / four-wave
    wino3_kernel) on the same operands: the same products in the same K order per accumulator, so the two agree to
    a few ulp of the accumulated magnitude; input read through a strided view (ldx > Cin)."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_fused_kernel
    lib = _lib.load()
    n, h, w, ci, co, xpad = case
    rng = np.random.default_rng(77 + ci + co)
    ldx = ci + xpad
    xs = np.full((n, h, w, ldx), 1e3, np.float32)        # the channels beyond Cin must never be read
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    xs[..., :ci] = x
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    dx, du, db = dev(xs), dev(pack_winograd_fused_kernel(wt)), dev(b)
    outs = []
    for form in (0, 1):
        d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ldx, Cout=co, cout_pad=-(-co // 32) * 32, ldy=co, kh=3, kw=3,
                          stride=1, relu=1, wino_form=form)
        y = torch.full((n * h * w, co), -5.0, device='cuda')
        _lib.check(lib.kfn_conv2d_winograd_fused(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(),
                                                 stream()), 'fused')
        sync()
        outs.append(y.cpu().numpy())
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 1, True)
    for o in outs:
        _check_err(np.abs(o.reshape(ref.shape) - ref).max(), x, wt, 'f23', 'wino fused forms %s' % (case,))
    assert np.abs(outs[0] - outs[1]).max() <= 1e-5 * max(1.0, np.abs(ref).max())


# kfn_conv2d_winograd_s2 (3x3 stride 2, even images): blocks straddling images (Th % 4 != 0), odd output sizes
# where the last 2x2 tile overhangs, ragged tile blocks and channel tiles, strided output, many super-steps
S2_CASES = [(1, 16, 16, 16, 128), (2, 14, 18, 32, 160), (1, 120, 160, 64, 128), (3, 30, 34, 48, 36),
            (5, 20, 24, 16, 8), (2, 60, 80, 256, 256), (17, 14, 16, 32, 136), (1, 64, 96, 512, 128)]


@pytest.mark.parametrize('form', [0, 4])      # four waves (wino_s2_kernel), KFN_WINO_FORM_S2_EIGHT_WAVE (wino_s2b_kernel)
@pytest.mark.parametrize('relu', [1, 0])
@pytest.mark.parametrize('case', S2_CASES)
def test_winograd_s2_vs_oracle(case, relu, form):
    """kfn_conv2d_winograd_s2 (polyphase + F(2,2): 25 MFMA streams into 9 accumulators per 2x2 outputs; four waves on 32x32x2
    MFMA tiles or eight on 16x16x4) == the oracle's stride-2 SAME convolution up to fp32 round-off; strided output window
    (ldy = Cout + 8: the 16-byte store path when Cout % 4 == 0, the dword path otherwise), guard rows untouched."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_s2_kernel, pack_winograd_s2_kernel_b
    lib = _lib.load()
    n, h, w, ci, co = case
    ho, wo = h // 2, w // 2
    rng = np.random.default_rng(n * 1000 + h * 10 + ci + 9)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ldy = co + 8
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=ldy, kh=3, kw=3,
                      stride=2, relu=relu, wino_form=form)
    assert lib.kfn_winograd_s2_supported(C.byref(d)) == 1
    GUARD = 64
    y = torch.full((n * ho * wo + GUARD, ldy), -5.0, device='cuda')
    dx, du, db = dev(x), dev((pack_winograd_s2_kernel_b if form == 4 else pack_winograd_s2_kernel)(wt)), dev(b)
    _lib.check(lib.kfn_conv2d_winograd_s2(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(),
                                          stream()), 'wino_s2')
    sync()
    got = y.cpu().numpy()
    assert np.all(got[:, co:] == -5.0) and np.all(got[n * ho * wo:] == -5.0)
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 2, bool(relu))
    assert ref.shape == (n, ho, wo, co)
    err = np.abs(got[:n * ho * wo, :co].reshape(ref.shape) - ref).max()
    _check_err(err, x, wt, 'f22s2', 'wino s2 %s relu %d form %d' % (case, relu, form))


# the F(4,2) form (wino_s2c_kernel, KFN_WINO_FORM_S2_F42): H, W multiples of 8, H >= 32; blocks straddling images (Th % 4 != 0),
# ragged tile-block columns and channel tiles, many super-steps, SCoordNet's three layers at one frame
S2C_CASES = [(1, 32, 32, 16, 128), (2, 40, 48, 32, 160), (1, 120, 160, 64, 128), (3, 32, 40, 48, 36), (5, 32, 32, 16, 8),
             (2, 64, 80, 256, 256), (7, 40, 32, 32, 136), (1, 64, 96, 512, 128), (1, 120, 160, 512, 1024),
             # the PERSISTENT form (wino_s2c_pkernel: at least two workgroups per CU, 2 .. 8 super-steps): workgroups walking 2-3 tile
             # blocks each, blocks straddling images (Th = 30), a ragged last round, channel groups that change along a walk
             (8, 120, 160, 32, 512), (4, 240, 320, 64, 256), (10, 120, 160, 128, 384)]


@pytest.mark.parametrize('relu', [1, 0])
@pytest.mark.parametrize('case', S2C_CASES)
def test_winograd_s2_f42_vs_oracle(case, relu):
    """kfn_conv2d_winograd_s2 with wino_form = KFN_WINO_FORM_S2_F42 (polyphase + F(4,2) on 4x4 output tiles: 81 products into 25
    accumulators per 16 outputs) == the oracle's stride-2 SAME convolution up to fp32 round-off (error class 'f42s2': the kernel
    measures up to 203 units of sqrt(1 + K/256) eps32 S at Cin = 512 -- direct 45, F(4x4,3x3) 526 -- allowance 400); strided output window,
    guard rows untouched; kfn_winograd_s2_supported answers for the form."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_s2_kernel_c
    lib = _lib.load()
    n, h, w, ci, co = case
    ho, wo = h // 2, w // 2
    rng = np.random.default_rng(n * 1000 + h * 10 + ci + 19)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ldy = co + 8
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=ldy, kh=3, kw=3,
                      stride=2, relu=relu, wino_form=_lib.WINO_FORM_S2_F42)
    assert lib.kfn_winograd_s2_supported(C.byref(d)) == 1
    GUARD = 64
    y = torch.full((n * ho * wo + GUARD, ldy), -5.0, device='cuda')
    dx, du, db = dev(x), dev(pack_winograd_s2_kernel_c(wt)), dev(b)
    _lib.check(lib.kfn_conv2d_winograd_s2(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(), stream()), 'wino_s2c')
    sync()
    got = y.cpu().numpy()
    assert np.all(got[:, co:] == -5.0) and np.all(got[n * ho * wo:] == -5.0)
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 2, bool(relu))
    err = np.abs(got[:n * ho * wo, :co].reshape(ref.shape) - ref).max()
    _check_err(err, x, wt, 'f42s2', 'wino s2c %s relu %d' % (case, relu))


# channel-blocked activations (KFN_LAYOUT_C16, round 6): (kind, form, N, H, W, Cin, Cout) -- whole and ragged blocks, blocks straddling
# images, one and many super-steps, the persistent stride-2 form, Cout of 16 / several channel groups
C16_CASES = [('f43', 3, 2, 32, 16, 16, 64), ('f43', 3, 3, 60, 80, 32, 64), ('f43', 3, 5, 36, 20, 16, 96), ('f43', 3, 3, 29, 35, 32, 112),
             ('f43', 3, 9, 32, 16, 512, 64), ('f43', 3, 2, 60, 80, 256, 128), ('f43', 3, 1, 29, 12, 32, 16),
             ('s2', 5, 1, 32, 32, 16, 128), ('s2', 5, 2, 40, 48, 32, 160), ('s2', 5, 3, 32, 40, 48, 48), ('s2', 5, 7, 40, 32, 32, 144),
             ('s2', 5, 1, 64, 96, 512, 128), ('s2', 5, 8, 120, 160, 32, 512), ('s2', 5, 10, 120, 160, 128, 384)]


@pytest.mark.parametrize('layouts', [('c16', 'c16'), ('c16', 'nhwc'), ('nhwc', 'c16')])
@pytest.mark.parametrize('case', C16_CASES)
def test_winograd_channel_blocked_layout(case, layouts):
    """kfn_conv_desc.x_layout / y_layout = KFN_LAYOUT_C16 (per image [C/16][H][W][16]) on the two kernels that take it -- wino4b_kernel
    and wino_s2c_kernel / wino_s2c_pkernel: blocked in, blocked out, and each mixed with NHWC (the ends of SCoordNet's blocked chain:
    conv1b reads NHWC, conv6 writes NHWC).  The layout only moves addresses: every combination must be BIT-IDENTICAL to the NHWC
    launch of the same operands (same products, same order), which the tests above hold to the oracle; the oracle is compared too."""
    from tests.gpu_util import run_winograd
    kind, form, n, h, w, ci, co = case
    rng = np.random.default_rng(n * 1000 + h * 10 + ci + 61)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    base = run_winograd(kind, x, wt, b, relu=True, form=form)
    got = run_winograd(kind, x, wt, b, relu=True, form=form, x_layout=layouts[0], y_layout=layouts[1])
    assert np.array_equal(got, base), 'layouts %s: max |diff| %g' % (layouts, np.abs(got - base).max())
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 2 if kind == 's2' else 1, True)
    _check_err(np.abs(got - ref).max(), x, wt, 'f42s2' if kind == 's2' else 'f43', 'c16 %s %s' % (case, layouts))


def test_channel_blocked_layout_is_refused_where_no_kernel_reads_it():
    """Every entry point but the two above answers KFN_ERR_UNSUPPORTED for a non-zero layout (instead of reading the buffer as NHWC);
    the two refuse blocked tensors that are not dense / whose channel count is not a multiple of 16; unknown layout values are
    KFN_ERR_ARG.  Nothing is launched."""
    import torch
    from kfnet_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(1 << 18, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    p = buf.data_ptr()
    ok = dict(N=1, H=32, W=32, Cin=32, ldx=32, Cout=64, cout_pad=64, ldy=64, kh=3, kw=3, stride=1)
    blk = dict(x_layout=_lib.LAYOUT_C16, y_layout=_lib.LAYOUT_C16)
    d = _lib.ConvDesc(**dict(ok, **blk))
    assert lib.kfn_conv2d_nhwc(C.byref(d), p, p, None, p, st) == -3 and b'NHWC only' in lib.kfn_last_error()
    assert lib.kfn_conv2d_winograd_fused(C.byref(d), p, p, None, p, st) == -3
    assert lib.kfn_conv2d_winograd_f43(C.byref(_lib.ConvDesc(**dict(ok, wino_form=2, **blk))), p, p, None, p, st) == -3     # four-wave form
    assert lib.kfn_winograd_f43_supported(C.byref(_lib.ConvDesc(**dict(ok, wino_form=2, **blk)))) == 0
    assert lib.kfn_winograd_f43_supported(C.byref(_lib.ConvDesc(**dict(ok, wino_form=3, **blk)))) == 1
    assert lib.kfn_conv2d_winograd_f43_splitk(C.byref(_lib.ConvDesc(**dict(ok, wino_form=3, **blk))), p, p, None, p, p, 2, st) == -3
    s2 = dict(ok, stride=2)
    assert lib.kfn_conv2d_winograd_s2(C.byref(_lib.ConvDesc(**dict(s2, wino_form=4, **blk))), p, p, None, p, st) == -3
    assert lib.kfn_winograd_s2_supported(C.byref(_lib.ConvDesc(**dict(s2, wino_form=4, **blk)))) == 0
    assert lib.kfn_winograd_s2_supported(C.byref(_lib.ConvDesc(**dict(s2, wino_form=5, **blk)))) == 1
    for form, fn, sup in ((3, lib.kfn_conv2d_winograd_f43, lib.kfn_winograd_f43_supported), (5, lib.kfn_conv2d_winograd_s2, lib.kfn_winograd_s2_supported)):
        base = dict(ok, stride=1 if form == 3 else 2, wino_form=form)
        for bad in (dict(ldx=40, x_layout=1), dict(ldy=72, y_layout=1), dict(Cout=40, cout_pad=64, ldy=40, y_layout=1)):
            dd = _lib.ConvDesc(**dict(base, **bad))
            assert sup(C.byref(dd)) == 0, (form, bad)
            assert fn(C.byref(dd), p, p, None, p, st) == -3, (form, bad)
        assert fn(C.byref(_lib.ConvDesc(**dict(base, x_layout=2))), p, p, None, p, st) == -1


def test_winograd_s2_f42_refuses_what_it_cannot_take():
    """H or W not a multiple of 8, H < 32, Cin % 16, Cout % 4, ldy % 4: not supported -> KFN_ERR_UNSUPPORTED with a message, nothing
    launched (the graph routes such layers to the F(2,2) form)."""
    import torch
    from kfnet_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(1 << 16, device='cuda')
    for kw_ in [dict(H=36, W=32), dict(H=32, W=36), dict(H=24, W=32), dict(Cin=24), dict(Cout=6), dict(ldy=130)]:
        a = dict(N=1, H=32, W=32, Cin=16, ldx=16, Cout=8, cout_pad=32, ldy=8, kh=3, kw=3, stride=2, relu=0, wino_form=_lib.WINO_FORM_S2_F42)
        a.update(kw_)
        if 'Cin' in kw_:
            a['ldx'] = kw_['Cin']
        d = _lib.ConvDesc(**a)
        assert lib.kfn_winograd_s2_supported(C.byref(d)) == 0, kw_
        rc = lib.kfn_conv2d_winograd_s2(C.byref(d), buf.data_ptr(), buf.data_ptr(), None, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == -3 and b'F(4,2)' in lib.kfn_last_error(), kw_


@pytest.mark.parametrize('case', [(1, 16, 16, 16, 128), (2, 14, 18, 32, 160), (1, 120, 160, 64, 128), (2, 60, 80, 256, 256),
                                  (17, 14, 16, 32, 136)])
def test_winograd_s2_fp16_operands(case):
    """kfn_conv2d_winograd_s2 with operand_dtype F16 (BASELINE config 5): transform in fp32, V and the weight
    fragments rounded to fp16, fp16 MFMAs with fp32 accumulation -- within ~1 % of the output range of the exact
    convolution, and measurably not the fp32 path."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import as_f16, pack_winograd_s2_kernel
    lib = _lib.load()
    n, h, w, ci, co = case
    rng = np.random.default_rng(n * 1000 + h * 10 + ci + 12)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=co, kh=3, kw=3,
                      stride=2, relu=1, operand_dtype=_lib.OPERAND_F16)
    assert lib.kfn_winograd_s2_supported(C.byref(d)) == 1
    y = torch.zeros((n * (h // 2) * (w // 2), co), device='cuda')
    dx, du, db = dev(x), dev(as_f16(pack_winograd_s2_kernel)(wt)), dev(b)
    _lib.check(lib.kfn_conv2d_winograd_s2(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(), stream()),
               'wino_s2 f16')
    sync()
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 2, True)
    err = np.abs(y.cpu().numpy().reshape(ref.shape) - ref).max()
    scale = np.abs(ref).max()
    print('wino_s2 f16: err %.3g, scale %.3g' % (err, scale))
    assert 0 < err <= 1e-2 * max(1.0, scale), err


def test_winograd_s2_strided_input_and_unsupported_shapes():
    """Input inside a wider buffer (ldx > Cin) and the shapes the polyphase kernel declines (odd sizes -- TF then
    pads before the image too --, stride 1, Cin % 16, too few tile rows)."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_s2_kernel
    lib = _lib.load()
    rng = np.random.default_rng(78)
    n, h, w, ci, co, ldx, off = 2, 16, 20, 32, 128, 48, 8
    x = rng.normal(size=(n, h, w, ci)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    xb = np.full((n * h * w, ldx), 9.0, np.float32)
    xb[:, off:off + ci] = x.reshape(-1, ci)
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ldx, Cout=co, cout_pad=128, ldy=co, kh=3, kw=3, stride=2, relu=0)
    dxb = dev(xb)
    y = torch.zeros((n * (h // 2) * (w // 2), co), device='cuda')
    _lib.check(lib.kfn_conv2d_winograd_s2(C.byref(d), dxb.data_ptr() + off * 4, dev(pack_winograd_s2_kernel(wt)).data_ptr(),
                                          None, y.data_ptr(), stream()), 'wino_s2 strided')
    sync()
    ref = O.conv2d_same(x.astype(np.float64), wt, None, 2, False)
    _check_err(np.abs(y.cpu().numpy().reshape(ref.shape) - ref).max(), x, wt, 'f22s2', 'wino s2 strided input')
    for bad in (dict(H=15), dict(W=19), dict(stride=1), dict(Cin=24, ldx=24), dict(H=12)):
        kw = dict(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=128, ldy=co, kh=3, kw=3, stride=2, relu=0)
        kw.update(bad)
        db_ = _lib.ConvDesc(**kw)
        assert lib.kfn_winograd_s2_supported(C.byref(db_)) == 0
        rc = lib.kfn_conv2d_winograd_s2(C.byref(db_), dxb.data_ptr(), dxb.data_ptr(), None, y.data_ptr(), stream())
        assert rc in (-3, -1), rc     # KFN_ERR_UNSUPPORTED / KFN_ERR_ARG


@pytest.mark.parametrize('case', [(1, 8, 8, 128, 128), (2, 7, 9, 128, 160), (1, 60, 80, 256, 128), (5, 30, 40, 64, 256),
                                  (17, 10, 12, 64, 136), (2, 12, 16, 512, 128)])
def test_winograd_fused_fp16_operands(case):
    """kfn_conv2d_winograd_fused with operand_dtype F16 (BASELINE config 5): V = B^T d B is formed in fp32 and
    rounded to fp16 when shared, U is fp16, the products are fp16 MFMAs with fp32 accumulation.  Against the
    fp64 convolution of the fp16-ROUNDED weights the error is that of rounding V (and of summing); against the
    exact one it must stay inside the fp16-operand tolerance of the direct kernel's test, x2 for the transform."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import as_f16, pack_winograd_fused_kernel
    lib = _lib.load()
    n, h, w, ci, co = case
    rng = np.random.default_rng(n * 1000 + h * 10 + ci + 6)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=co, kh=3, kw=3,
                      stride=1, relu=1, operand_dtype=_lib.OPERAND_F16)
    assert lib.kfn_winograd_fused_supported(C.byref(d)) == 1
    y = torch.zeros((n * h * w, co), device='cuda')
    dx, du, db = dev(x), dev(as_f16(pack_winograd_fused_kernel)(wt)), dev(b)
    _lib.check(lib.kfn_conv2d_winograd_fused(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(),
                                             stream()), 'wino3 f16')
    sync()
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 1, True)
    err = np.abs(y.cpu().numpy().reshape(ref.shape) - ref).max()
    scale = np.abs(ref).max()
    print('wino3 f16: err %.3g, scale %.3g' % (err, scale))
    assert 0 < err <= 1e-2 * max(1.0, scale), err        # reduced precision, and no worse than ~1 % of the output range
    # and the shapes the fp16 path declines (one-wave form): told, not mis-computed
    d2 = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=64, cout_pad=64, ldy=64, kh=3, kw=3, stride=1, relu=1,
                       operand_dtype=_lib.OPERAND_F16)
    assert lib.kfn_winograd_fused_supported(C.byref(d2)) == 0
    assert lib.kfn_conv2d_winograd_fused(C.byref(d2), dx.data_ptr(), du.data_ptr(), None, y.data_ptr(), stream()) == -3


def test_winograd_fused_strided_input_and_unsupported_shapes():
    """Input living in a wider buffer (ldx > Cin, the concat case) and the shapes the single-kernel path
    declines (Cin % 16, fewer than 4 tile rows): the host must be told, not handed wrong numbers."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_fused_kernel
    lib = _lib.load()
    rng = np.random.default_rng(77)
    n, h, w, ci, co, ldx, off = 2, 12, 20, 32, 64, 48, 8
    x = rng.normal(size=(n, h, w, ci)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    xb = np.full((n * h * w, ldx), 9.0, np.float32)
    xb[:, off:off + ci] = x.reshape(-1, ci)
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ldx, Cout=co, cout_pad=64, ldy=co, kh=3, kw=3, stride=1, relu=0)
    y = torch.zeros((n * h * w, co), device='cuda')
    dx, du = dev(xb), dev(pack_winograd_fused_kernel(wt))
    _lib.check(lib.kfn_conv2d_winograd_fused(C.byref(d), dx.data_ptr() + 4 * off, du.data_ptr(), None, y.data_ptr(),
                                             stream()), 'wino2')
    sync()
    ref = O.conv2d_same(x.astype(np.float64), wt, None, 1, False)
    _check_err(np.abs(y.cpu().numpy().reshape(ref.shape) - ref).max(), x, wt, 'f23', 'wino fused strided input')
    for bad in (dict(Cin=8, ldx=8), dict(H=5), dict(Cin=24, ldx=24)):
        kw = dict(N=1, H=16, W=16, Cin=32, ldx=32, Cout=32, cout_pad=32, ldy=32, kh=3, kw=3, stride=1)
        kw.update(bad)
        db = _lib.ConvDesc(**kw)
        assert lib.kfn_winograd_fused_supported(C.byref(db)) == 0
        assert lib.kfn_conv2d_winograd_fused(C.byref(db), dx.data_ptr(), du.data_ptr(), None, y.data_ptr(), stream()) == -3


@pytest.mark.parametrize('shape', [(2, 7, 9, 32, 32), (1, 60, 80, 32, 32), (3, 5, 4, 16, 48), (2, 3, 2, 32, 32)])
def test_cost_volume_factored_vs_oracle(shape):
    """pad(f1) -> 3x3 conv with the 9 class kernels, 1x1 conv of f2 with the class sums,
    kfn_cost_volume_gather == conv0(BuildCoordVolume(f1, f2)) of the oracle (same borders:
    window-border SAME padding of conv0 and zero fill of the shifted f1 outside the image)."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_cvol_bias, pack_cvol_g_kernel, pack_cvol_t_kernel
    lib = _lib.load()
    N, H, W, Cc, co = shape
    rng = np.random.default_rng(78)
    f = rng.normal(size=(N + 1, H, W, Cc)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, Cc, co)) / np.sqrt(9 * Cc)).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    c9 = 9 * co
    fd = dev(f)
    wg, wtt, b9 = dev(pack_cvol_g_kernel(wt)), dev(pack_cvol_t_kernel(wt)), dev(pack_cvol_bias(b))
    f1p = torch.empty(N * (H + 4) * (W + 4) * Cc, device='cuda')
    Gp = torch.empty(N * (H + 4) * (W + 4) * c9, device='cuda')
    T = torch.empty(N * H * W * c9, device='cuda')
    ldy, off = co + 16, 16
    y = torch.full((N * H * W * 64, ldy), -9.0, device='cuda')
    st = stream()
    _lib.check(lib.kfn_pad_nhwc(fd.data_ptr(), f1p.data_ptr(), N, H, W, Cc, 2, st), 'pad')
    cp = -(-c9 // 32) * 32
    dg = _lib.ConvDesc(N=N, H=H + 4, W=W + 4, Cin=Cc, ldx=Cc, Cout=c9, cout_pad=cp, ldy=c9, kh=3, kw=3, stride=1)
    _lib.check(lib.kfn_conv2d_nhwc(C.byref(dg), f1p.data_ptr(), wg.data_ptr(), None, Gp.data_ptr(), st), 'G')
    dt = _lib.ConvDesc(N=N, H=H, W=W, Cin=Cc, ldx=Cc, Cout=c9, cout_pad=cp, ldy=c9, kh=1, kw=1, stride=1)
    _lib.check(lib.kfn_conv2d_nhwc(C.byref(dt), fd.data_ptr() + H * W * Cc * 4, wtt.data_ptr(), b9.data_ptr(),
                                   T.data_ptr(), st), 'T')
    _lib.check(lib.kfn_cost_volume_gather(T.data_ptr(), Gp.data_ptr(), y.data_ptr() + off * 4, N, H, W, co, ldy, 1, st),
               'gather')
    sync()
    got = y.cpu().numpy()
    assert np.all(got[:, :off] == -9.0)
    for n in range(N):
        vol, _ = O.coord_volume(f[n:n + 1].astype(np.float64), f[n + 1:n + 2].astype(np.float64), 8)
        ref = O.conv2d_same(vol, wt, b, 1, True).reshape(H * W * 64, co)
        g = got[n * H * W * 64:(n + 1) * H * W * 64, off:off + co]
        assert np.abs(g - ref).max() < 2e-5


def _factored_maps(f, wt, b, N, H, W):
    """T [N,H,W,288] and Gp [N,H+4,W+4,288] of the factored cost volume (as the graph computes them: kfn_pad_nhwc +
    two kfn_conv2d_nhwc launches with the class kernels) for frame pairs (f[n], f[n+1])."""
    import torch
    from tests.gpu_util import dev, stream
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_cvol_bias, pack_cvol_g_kernel, pack_cvol_t_kernel
    lib = _lib.load()
    Cc, co, c9 = 32, 32, 288
    fd = dev(f)
    wg, wtt, b9 = dev(pack_cvol_g_kernel(wt)), dev(pack_cvol_t_kernel(wt)), dev(pack_cvol_bias(b))
    f1p = torch.empty(N * (H + 4) * (W + 4) * Cc, device='cuda')
    Gp = torch.empty(N * (H + 4) * (W + 4) * c9, device='cuda')
    T = torch.empty(N * H * W * c9, device='cuda')
    st = stream()
    _lib.check(lib.kfn_pad_nhwc(fd.data_ptr(), f1p.data_ptr(), N, H, W, Cc, 2, st), 'pad')
    dg = _lib.ConvDesc(N=N, H=H + 4, W=W + 4, Cin=Cc, ldx=Cc, Cout=c9, cout_pad=c9, ldy=c9, kh=3, kw=3, stride=1)
    _lib.check(lib.kfn_conv2d_nhwc(C.byref(dg), f1p.data_ptr(), wg.data_ptr(), None, Gp.data_ptr(), st), 'G')
    dt = _lib.ConvDesc(N=N, H=H, W=W, Cin=Cc, ldx=Cc, Cout=c9, cout_pad=c9, ldy=c9, kh=1, kw=1, stride=1)
    _lib.check(lib.kfn_conv2d_nhwc(C.byref(dt), fd.data_ptr() + H * W * Cc * 4, wtt.data_ptr(), b9.data_ptr(),
                                   T.data_ptr(), st), 'T')
    return T, Gp, (fd, wg, wtt, b9, f1p)


def _oracle_conv0(f, wt, b, N, H, W):
    """relu(conv0(BuildCoordVolume(f[n], f[n+1]))) for every pair: [N*H*W, 8, 8, 32] (fp64)."""
    out = []
    for n in range(N):
        vol, _ = O.coord_volume(f[n:n + 1].astype(np.float64), f[n + 1:n + 2].astype(np.float64), 8)
        out.append(O.conv2d_same(vol, wt, b, 1, True))
    return np.concatenate(out)


@pytest.mark.parametrize('shape', [(2, 7, 9), (1, 60, 80), (3, 5, 4), (2, 3, 2)])
def test_oflow_head_vs_oracle(shape):
    """kfn_oflow_head (conv0 from the factored maps + conv1a, one wave per window) == the oracle's
    conv2d(stride 2, relu) of relu(conv0(cost volume)): image borders (zero fill of the shifted f1), window borders
    (conv0's SAME padding) and conv1a's bottom / right SAME padding included."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_oflow_head_kernel
    lib = _lib.load()
    N, H, W = shape
    rng = np.random.default_rng(91)
    f = rng.normal(size=(N + 1, H, W, 32)).astype(np.float32)
    w0 = (rng.normal(size=(3, 3, 32, 32)) / np.sqrt(9 * 32)).astype(np.float32)
    b0 = rng.normal(size=32).astype(np.float32)
    w1 = (rng.normal(size=(3, 3, 32, 32)) * np.sqrt(2.0 / (9 * 32))).astype(np.float32)
    b1 = (rng.normal(size=32) * 0.1).astype(np.float32)
    T, Gp, keep = _factored_maps(f, w0, b0, N, H, W)
    P = N * H * W
    y = torch.full((P * 16 * 32 + 64,), -7.0, device='cuda')
    dw1, db1 = dev(pack_oflow_head_kernel(w1)), dev(b1)
    _lib.check(lib.kfn_oflow_head(T.data_ptr(), Gp.data_ptr(), N, H, W, 1, dw1.data_ptr(), db1.data_ptr(), y.data_ptr(),
                                  stream()), 'oflow_head')
    sync()
    got = y.cpu().numpy()
    assert np.all(got[P * 512:] == -7.0)
    ref = O.conv2d_same(_oracle_conv0(f, w0, b0, N, H, W), w1, b1, 2, True)          # [P,4,4,32]
    err = np.abs(got[:P * 512].reshape(ref.shape) - ref).max()
    assert err < 4e-5 * max(1.0, float(np.abs(ref).max())), err
    # BASELINE config 5's form: conv1a on fp16 MFMAs (operands rounded to halfs: 2^-11 relative each over K = 288; a wrong
    # fragment layout is an O(1) error), everything around it fp32
    from kfnet_amd.graph import pack_oflow_head_kernel_f16
    dw1h = dev(pack_oflow_head_kernel_f16(w1))
    assert dw1h.dtype == torch.float16 and tuple(dw1h.shape) == (36, 64, 4)
    y16 = torch.full((P * 16 * 32 + 64,), -7.0, device='cuda')
    _lib.check(lib.kfn_oflow_head_f16(T.data_ptr(), Gp.data_ptr(), N, H, W, 1, dw1h.data_ptr(), db1.data_ptr(),
                                      y16.data_ptr(), stream()), 'oflow_head_f16')
    sync()
    got16 = y16.cpu().numpy()
    assert np.all(got16[P * 512:] == -7.0)
    err16 = np.abs(got16[:P * 512].reshape(ref.shape) - ref).max()
    assert 0 < err16 < 4e-3 * max(1.0, float(np.abs(ref).max())), err16


@pytest.mark.parametrize('shape', [(2, 7, 9), (1, 60, 80), (3, 5, 4)])
def test_oflow_tail2_vs_oracle(shape):
    """kfn_oflow_tail2 (upconv0 of conv5's patch ++ recomputed conv0 -> conv6 -> prediction -> softmax -> soft-argmax)
    == the oracle's conv2d_transpose / concat / conv2d chain on the materialised tensors."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_oflow_tail_kernel, pack_oflow_upconv_kernel
    lib = _lib.load()
    N, H, W = shape
    P = N * H * W
    rng = np.random.default_rng(57)
    f = rng.normal(size=(N + 1, H, W, 32)).astype(np.float32)
    w0 = (rng.normal(size=(3, 3, 32, 32)) / np.sqrt(9 * 32)).astype(np.float32)
    b0 = rng.normal(size=32).astype(np.float32)
    x5 = np.maximum(rng.normal(size=(P, 4, 4, 32)), 0).astype(np.float32)
    wu = (rng.normal(size=(3, 3, 16, 32)) * np.sqrt(2.0 / (4 * 32))).astype(np.float32)     # [kh,kw,Cout,Cin]
    bu = (rng.normal(size=16) * 0.1).astype(np.float32)
    w6 = (rng.normal(size=(3, 3, 48, 16)) * np.sqrt(2.0 / (9 * 48))).astype(np.float32)
    b6 = (rng.normal(size=16) * 0.1).astype(np.float32)
    wp = (rng.normal(size=(3, 3, 16, 1)) * 0.5).astype(np.float32)
    bp = np.array([0.3], np.float32)
    T, Gp, keep = _factored_maps(f, w0, b0, N, H, W)
    flow = torch.zeros(P * 2, device='cuda')
    logits = torch.zeros(P * 64, device='cuda')
    d5, du, dbu, d6, db6, dwp, dbp = (dev(x5), dev(pack_oflow_upconv_kernel(wu)), dev(bu), dev(pack_oflow_tail_kernel(w6)),
                                      dev(b6), dev(wp[..., 0]), dev(bp))
    _lib.check(lib.kfn_oflow_tail2(T.data_ptr(), Gp.data_ptr(), N, H, W, 1, d5.data_ptr(), du.data_ptr(), dbu.data_ptr(),
                                   d6.data_ptr(), db6.data_ptr(), dwp.data_ptr(), dbp.data_ptr(), flow.data_ptr(),
                                   logits.data_ptr(), stream()), 'oflow_tail2')
    sync()
    up = O.conv2d_transpose_same(x5.astype(np.float64), wu, bu, 2, True)                    # [P,8,8,16]
    cat = np.concatenate([up, _oracle_conv0(f, w0, b0, N, H, W)], -1)                       # concat0 = [upconv0, conv0]
    mid = O.conv2d_same(cat, w6, b6, 1, True)
    ref_logits = O.conv2d_same(mid, wp, bp, 1, False)[..., 0].reshape(P, 64)
    pr = O.softmax(ref_logits)
    offs = O.coord_volume(np.zeros((1, 2, 2, 1)), np.zeros((1, 2, 2, 1)), 8)[1]
    tol = 1e-5 * max(1.0, float(np.abs(ref_logits).max()))
    assert np.abs(logits.cpu().numpy().reshape(P, 64) - ref_logits).max() < tol
    assert np.abs(flow.cpu().numpy().reshape(P, 2) - pr.dot(offs)).max() < 20 * tol
    flow2 = torch.zeros(P * 2, device='cuda')          # without the optional logits output
    _lib.check(lib.kfn_oflow_tail2(T.data_ptr(), Gp.data_ptr(), N, H, W, 1, d5.data_ptr(), du.data_ptr(), dbu.data_ptr(),
                                   d6.data_ptr(), db6.data_ptr(), dwp.data_ptr(), dbp.data_ptr(), flow2.data_ptr(), None,
                                   stream()), 'oflow_tail2')
    sync()
    assert torch.equal(flow, flow2)
    # BASELINE config 5's form: upconv0 / conv6 on fp16 MFMAs (operands rounded to halfs, fp32 accumulation and everything
    # around them fp32).  Tolerance: 2^-11 relative per operand over K = 432 products of O(1) logits -- a wrong fragment
    # layout is an O(1) error
    from kfnet_amd.graph import pack_oflow_tail_kernel_f16, pack_oflow_upconv_kernel_f16
    duh, d6h = dev(pack_oflow_upconv_kernel_f16(wu)), dev(pack_oflow_tail_kernel_f16(w6))
    assert duh.dtype == torch.float16 and tuple(duh.shape) == (18, 64, 4) and tuple(d6h.shape) == (27, 64, 4)
    flow3 = torch.zeros(P * 2, device='cuda')
    logits3 = torch.zeros(P * 64, device='cuda')
    _lib.check(lib.kfn_oflow_tail2_f16(T.data_ptr(), Gp.data_ptr(), N, H, W, 1, d5.data_ptr(), duh.data_ptr(), dbu.data_ptr(),
                                       d6h.data_ptr(), db6.data_ptr(), dwp.data_ptr(), dbp.data_ptr(), flow3.data_ptr(),
                                       logits3.data_ptr(), stream()), 'oflow_tail2_f16')
    sync()
    tol16 = 4e-3 * max(1.0, float(np.abs(ref_logits).max()))
    e_l = np.abs(logits3.cpu().numpy().reshape(P, 64) - ref_logits).max()
    e_f = np.abs(flow3.cpu().numpy().reshape(P, 2) - pr.dot(offs)).max()
    assert 0 < e_l < tol16 and e_f < 20 * tol16, (e_l, e_f, tol16)


# ---- border handling of the minimal-filtering kernels, pinned exactly (Next #3) ---------------------------------------
# TF 'SAME' semantics (cnn_wrapper/network.py:116-135): zero padding (1,1) at stride 1, (0,1) at stride 2 on even
# sizes.  The random-data cases above bound the round-off; these cases make a wrong or dropped tap at ONE border position an O(1)
# error.  (kind, form, shape (N,H,W,Cin,Cout), relative tolerance of the kernel's arithmetic on exactly representable data: the
# F(2,2)-based kernels are EXACT on small integers -- G g G^T has quarters, B^T d B sums of four integers -- and get a token
# 2e-6; F(4x4,3x3)'s G holds sixths, so U is rounded: a CPU emulation of the fp32 evaluation measures 3.6-5.0e-5 absolute on the
# selector case and 1.9-3.3e-5 on the impulse case, tools/experiments/conv_error_model.py's emulation on these very inputs)
BORDER_KERNELS = [
    ('fused', 1, (2, 14, 18, 32, 32), 2e-6),     # wino2_kernel (one wave): block of 8x4 tiles straddles the two images
    ('fused', 0, (2, 14, 18, 32, 64), 2e-6),     # wino3_pair_kernel
    ('fused', 0, (2, 14, 18, 32, 128), 2e-6),    # wino3_kernel (four waves)
    ('fused', 0, (1, 9, 70, 16, 32), 2e-6),      # odd sizes: the last 2x2 tile overhangs; ragged tile-block columns
    ('wino', 0, (2, 7, 9, 32, 36), 2e-6),        # two-kernel form, odd sizes
    ('f43', 2, (2, 36, 40, 16, 64), 4e-5),       # wino4_kernel: Th = 9 -> 8-row tile blocks straddle the images, Tw = 10 ragged
    ('f43', 3, (2, 36, 40, 16, 64), 4e-5),       # wino4b_kernel
    ('f43', 3, (1, 29, 35, 32, 100), 4e-5),      # sizes not multiples of 4: the last 4x4 tiles overhang; Cout % 64 != 0
    ('f43', 2, (1, 29, 35, 32, 100), 4e-5),
    ('s2', 0, (2, 14, 18, 32, 160), 2e-6),       # wino_s2_kernel: odd output sizes (7x9), pad (0,1)
    ('s2', 4, (2, 14, 18, 32, 160), 2e-6),       # wino_s2b_kernel
    ('s2', 4, (1, 30, 34, 48, 36), 2e-6),        # dword-store path (ldy = Cout + 8 with Cout % 4 == 0 still wide; ragged channels)
    # wino_s2c_kernel (polyphase + F(4,2)): G holds sixths like F(4x4,3x3)'s, so U is rounded (a CPU emulation of the fp32
    # evaluation on these inputs: 1.1e-5 / 0.6e-5 absolute).  Th = 5: the 4-row tile blocks straddle the two images; Tw = 6 ragged
    ('s2', 5, (2, 40, 48, 32, 160), 4e-5),
    ('s2', 5, (1, 32, 72, 48, 36), 4e-5),        # ragged channels (Cout = 36 of 128), Tw = 9
]


@pytest.mark.parametrize('kind,form,shape,rtol', BORDER_KERNELS)
def test_winograd_border_taps_selector_weights(kind, form, shape, rtol):
    """Tap-selector weights: output channel t = 3a + b copies ONE input channel through the single tap (a, b) with weight 1,
    every other output channel has all-zero weights.  So y[..., t] must be the input channel SHIFTED by that tap with zeros
    shifted in from outside the image -- at every pixel, every border included -- and the silent channels must be exactly the
    bias.  Integer-valued data in [-8, 8] make every product and the direct sum exact; what remains is the round-off of the
    transforms on exactly representable data (rtol x the data range)."""
    from tests.gpu_util import run_winograd
    n, h, w, ci, co = shape
    rng = np.random.default_rng(1000 * h + w + ci)
    x = rng.integers(-8, 9, size=(n, h, w, ci)).astype(np.float32)
    wt = np.zeros((3, 3, ci, co), np.float32)
    src = [(5 * t + 3) % ci for t in range(9)]            # nine different input channels, in different 8-channel chunks
    for t in range(9):
        wt[t // 3, t % 3, src[t], t] = 1.0
    b = (np.arange(co) % 5 - 2).astype(np.float32)
    y = run_winograd(kind, x, wt, b, relu=False, form=form, ldx_pad=8 if kind == 'f43' else 0)
    stride = 2 if kind == 's2' else 1
    ref = O.conv2d_same(x.astype(np.float64), wt, b, stride, False)
    assert y.shape == ref.shape
    # the oracle's own statement of the case, independent of its convolution: a shifted copy with zero fill
    ho, wo = ref.shape[1:3]
    pad_h = max((ho - 1) * stride + 3 - h, 0) // 2
    pad_w = max((wo - 1) * stride + 3 - w, 0) // 2
    for t in range(9):
        a, bb = t // 3, t % 3
        exp = np.zeros((n, ho, wo))
        for oy in range(ho):
            iy = oy * stride + a - pad_h
            if not 0 <= iy < h:
                continue
            for ox in range(wo):
                ix = ox * stride + bb - pad_w
                if 0 <= ix < w:
                    exp[:, oy, ox] = x[:, iy, ix, src[t]]
        assert np.array_equal(ref[..., t] - b[t], exp), 'oracle disagrees with the shift statement of tap %d' % t
    err = np.abs(y - ref)
    tol = rtol * 8.0
    bad = np.argwhere(err > tol)
    assert bad.size == 0, ('%s form %d: %d outputs off by more than %.1e, first at (n,y,x,c) = %s: got %r want %r'
                           % (kind, form, len(bad), tol, tuple(bad[0]), y[tuple(bad[0])], ref[tuple(bad[0])]))
    # silent channels: nothing but the bias (a kernel that leaks a neighbouring channel's accumulator shows up here)
    assert np.abs(y[..., 9:] - b[9:]).max() <= tol


@pytest.mark.parametrize('kind,form,shape,rtol', BORDER_KERNELS)
def test_winograd_border_impulses_dense_weights(kind, form, shape, rtol):
    """The transposed view: single non-zero input pixels -- one at each corner, each edge, across the seam of the two
    images of the batch (last row of image 0 / first row of image 1: the kernels pack the tile rows of a batch), on tile-block
    seams and in the interior, each in its own input channel -- against dense integer weights: every impulse must scatter
    exactly the 3x3 (stride 2: the taps its parity admits) pattern of its channel's kernel, clipped at the image border, and
    nothing else anywhere (the support pattern of the direct convolution)."""
    from tests.gpu_util import run_winograd
    n, h, w, ci, co = shape
    rng = np.random.default_rng(7 * h + w + co)
    wt = rng.integers(-4, 5, size=(3, 3, ci, co)).astype(np.float32)
    x = np.zeros((n, h, w, ci), np.float32)
    pts = [(0, 0, 0), (0, 0, w - 1), (0, h - 1, 0), (n - 1, h - 1, w - 1), (0, 0, w // 2), (0, h // 2, 0),
           (0, h // 2, w - 1), (0, h - 1, w // 2), (n - 1, 0, w // 2 + 1), (0, h // 2, w // 2),
           (0, min(15, h - 2), min(15, w - 2)), (n - 1, min(16, h - 1), min(16, w - 1)), (0, min(31, h - 3), 3), (n - 1, min(8, h - 1), min(32, w - 3))]
    for i, (b_, yy, xx) in enumerate(pts):
        x[b_, yy, xx, i % ci] += float(1 + i % 3)
    y = run_winograd(kind, x, wt, None, relu=False, form=form, ldx_pad=8 if kind == 'f43' else 0)
    stride = 2 if kind == 's2' else 1
    ref = O.conv2d_same(x.astype(np.float64), wt, None, stride, False)
    tol = rtol * 3.0 * 4.0 * 4.0         # data range: amplitude <= 3, |w| <= 4, a few overlapping supports
    err = np.abs(y - ref)
    bad = np.argwhere(err > tol)
    assert bad.size == 0, ('%s form %d: %d outputs off by more than %.1e, first at (n,y,x,c) = %s: got %r want %r'
                           % (kind, form, len(bad), tol, tuple(bad[0]), y[tuple(bad[0])], ref[tuple(bad[0])]))
    # the support pattern: where the direct convolution is exactly zero the kernel's output is zero to the same tolerance and
    # the non-zero outputs carry the integer weights
    assert np.abs(y[ref == 0]).max() <= tol
    assert np.abs(ref).max() >= 4.0


# ---- split-K form of the F(4x4,3x3) kernel (BASELINE configs[1]: launches of fewer workgroups than CUs) ---------------------------
@pytest.mark.parametrize('case,k_split', [((1, 60, 80, 1024, 128), 3), ((1, 60, 80, 512, 256), 6), ((2, 32, 16, 64, 64), 2),
                                          ((2, 32, 16, 64, 64), 4), ((3, 29, 35, 32, 100), 2), ((1, 60, 80, 1024, 128), 1)])
def test_winograd_f43_splitk_vs_oracle(case, k_split):
    """kfn_conv2d_winograd_f43_splitk: k_split copies of the tile grid accumulate disjoint runs of input channels into the planes
    of a workspace, a second kernel adds the planes in a fixed order + bias + ReLU.  == the oracle within the F(4x4,3x3) bound;
    two runs are BIT-identical (no atomics, fixed order); the unsplit launch differs by summation order only; strided output
    (ldy = Cout + 8), guard rows and the columns behind Cout untouched."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_f43_kernel_b
    lib = _lib.load()
    n, h, w, ci, co = case
    rng = np.random.default_rng(n * 1000 + h * 10 + ci + 143)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ldy = co + 8
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=ldy, kh=3, kw=3, stride=1, relu=1,
                      wino_form=3)
    nb = C.c_size_t()
    _lib.check(lib.kfn_winograd_f43_splitk_workspace_bytes(C.byref(d), k_split, C.byref(nb)), 'ws bytes')
    assert nb.value == (k_split * n * h * w * co * 4 if k_split > 1 else 0)
    GUARD = 64
    ws = torch.full((nb.value // 4 + GUARD,), -7.0, device='cuda')
    dx, du, db = dev(x), dev(pack_winograd_f43_kernel_b(wt)), dev(np.concatenate([b, np.zeros((-co) % 4, np.float32)]))
    outs = []
    for rep in range(2):
        y = torch.full((n * h * w + GUARD, ldy), -5.0, device='cuda')
        _lib.check(lib.kfn_conv2d_winograd_f43_splitk(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(),
                                                      ws.data_ptr(), k_split, stream()), 'f43 split-K')
        sync()
        got = y.cpu().numpy()
        assert np.all(got[:, co:] == -5.0) and np.all(got[n * h * w:] == -5.0)
        outs.append(got[:n * h * w, :co].copy())
    assert bool((ws[nb.value // 4:] == -7.0).all()), 'the partial sums went past the workspace'
    assert np.array_equal(outs[0], outs[1]), 'split-K result changes from run to run'
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 1, True)
    _check_err(np.abs(outs[0].reshape(ref.shape) - ref).max(), x, wt, 'f43', 'wino f43 split-K %d %s' % (k_split, case))
    y1 = torch.full((n * h * w, ldy), -5.0, device='cuda')
    _lib.check(lib.kfn_conv2d_winograd_f43(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y1.data_ptr(), stream()), 'f43')
    sync()
    un = y1.cpu().numpy()[:, :co]
    assert np.abs(un - outs[0]).max() <= 2 * _conv_tol(x, wt, kind='f43')
    if k_split == 1:
        assert np.array_equal(un, outs[0])       # k_split = 1 IS the unsplit eight-wave launch


def test_winograd_f43_splitk_rejects_bad_splits():
    from kfnet_amd import _lib
    import torch
    lib = _lib.load()
    d = _lib.ConvDesc(N=1, H=32, W=32, Cin=64, ldx=64, Cout=64, cout_pad=64, ldy=64, kh=3, kw=3, stride=1, wino_form=3)
    buf = torch.zeros(1 << 20, device='cuda')
    args = lambda ks, ws=buf.data_ptr(): (C.byref(d), buf.data_ptr(), buf.data_ptr(), None, buf.data_ptr(), ws, ks, None)
    assert lib.kfn_conv2d_winograd_f43_splitk(*args(0)) != 0
    assert lib.kfn_conv2d_winograd_f43_splitk(*args(5)) != 0 and b'outside 1..4' in lib.kfn_last_error()   # Cin / 16 = 4 super-steps
    assert lib.kfn_conv2d_winograd_f43_splitk(*args(3)) != 0 and b'empty split' in lib.kfn_last_error()    # runs of 2: the third is empty
    assert lib.kfn_conv2d_winograd_f43_splitk(*args(2, None)) != 0 and b'workspace' in lib.kfn_last_error()
    d4 = _lib.ConvDesc(N=1, H=32, W=32, Cin=64, ldx=64, Cout=64, cout_pad=64, ldy=64, kh=3, kw=3, stride=1, wino_form=2)
    assert lib.kfn_conv2d_winograd_f43_splitk(C.byref(d4), buf.data_ptr(), buf.data_ptr(), None, buf.data_ptr(), buf.data_ptr(), 2, None) != 0


@pytest.mark.parametrize('case,k_split', [((1, 120, 160, 512, 1024), 4), ((2, 14, 18, 32, 160), 2), ((1, 64, 96, 64, 128), 2),
                                          ((1, 30, 34, 48, 36), 3), ((2, 60, 80, 256, 256), 1)])
def test_winograd_s2_splitk_vs_oracle(case, k_split):
    """kfn_conv2d_winograd_s2_splitk (eight-wave polyphase kernel, input channels cut into k_split runs, planes reduced in a
    fixed order): == the oracle's stride-2 SAME convolution within the polyphase bound, bit-identical from run to run, the
    unsplit launch differs by summation order only; strided output, guards untouched.  (1,120,160,512,1024) = conv4a at
    batch 1, the layer the split is for."""
    import torch
    from tests.gpu_util import dev, stream, sync
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_winograd_s2_kernel_b
    lib = _lib.load()
    n, h, w, ci, co = case
    ho, wo = h // 2, w // 2
    rng = np.random.default_rng(n * 1000 + h * 10 + ci + 29)
    x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ldy = co + 8
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=ldy, kh=3, kw=3, stride=2, relu=1,
                      wino_form=4)
    nb = C.c_size_t()
    _lib.check(lib.kfn_winograd_s2_splitk_workspace_bytes(C.byref(d), k_split, C.byref(nb)), 'ws bytes')
    assert nb.value == (k_split * n * ho * wo * co * 4 if k_split > 1 else 0)
    GUARD = 64
    ws = torch.full((nb.value // 4 + GUARD,), -7.0, device='cuda')
    dx, du, db = dev(x), dev(pack_winograd_s2_kernel_b(wt)), dev(b)
    outs = []
    for rep in range(2):
        y = torch.full((n * ho * wo + GUARD, ldy), -5.0, device='cuda')
        _lib.check(lib.kfn_conv2d_winograd_s2_splitk(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y.data_ptr(),
                                                     ws.data_ptr(), k_split, stream()), 's2 split-K')
        sync()
        got = y.cpu().numpy()
        assert np.all(got[:, co:] == -5.0) and np.all(got[n * ho * wo:] == -5.0)
        outs.append(got[:n * ho * wo, :co].copy())
    assert bool((ws[nb.value // 4:] == -7.0).all()), 'the partial sums went past the workspace'
    assert np.array_equal(outs[0], outs[1]), 'split-K result changes from run to run'
    ref = O.conv2d_same(x.astype(np.float64), wt, b, 2, True)
    _check_err(np.abs(outs[0].reshape(ref.shape) - ref).max(), x, wt, 'f22s2', 'wino s2 split-K %d %s' % (k_split, case))
    y1 = torch.full((n * ho * wo, ldy), -5.0, device='cuda')
    _lib.check(lib.kfn_conv2d_winograd_s2(C.byref(d), dx.data_ptr(), du.data_ptr(), db.data_ptr(), y1.data_ptr(), stream()), 's2')
    sync()
    un = y1.cpu().numpy()[:, :co]
    assert np.abs(un - outs[0]).max() <= 2 * _conv_tol(x, wt, kind='f22s2')
    if k_split == 1:
        assert np.array_equal(un, outs[0])
    # what the entry point refuses: a split with an empty run, a null workspace, the fp16 / four-wave forms
    if k_split > 1:
        n_super = ci // 16
        assert lib.kfn_conv2d_winograd_s2_splitk(C.byref(d), dx.data_ptr(), du.data_ptr(), None, y1.data_ptr(), ws.data_ptr(),
                                                 n_super + 1, stream()) != 0
        if n_super == 4:      # runs of 2 super-steps: a third run would be empty
            assert lib.kfn_conv2d_winograd_s2_splitk(C.byref(d), dx.data_ptr(), du.data_ptr(), None, y1.data_ptr(), ws.data_ptr(),
                                                     3, stream()) != 0 and b'empty split' in lib.kfn_last_error()
        assert lib.kfn_conv2d_winograd_s2_splitk(C.byref(d), dx.data_ptr(), du.data_ptr(), None, y1.data_ptr(), None, k_split, stream()) != 0
        d16 = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=ldy, kh=3, kw=3, stride=2,
                            operand_dtype=1)
        assert lib.kfn_conv2d_winograd_s2_splitk(C.byref(d16), dx.data_ptr(), du.data_ptr(), None, y1.data_ptr(), ws.data_ptr(), k_split, stream()) != 0
