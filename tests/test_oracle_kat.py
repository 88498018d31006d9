"""Known-answer tests for the CPU oracle, derived analytically from the cited reference
code.  These are the only pins the oracle has: the reference cannot
run here (no TensorFlow) and ships no golden vectors."""
import os

import numpy as np
import pytest

from oracle import kfnet_oracle as O
from oracle import kfnet_oracle_torch as OT


def test_kalman_equal_noise():  # KFNet/KFNet.py:154-160
    x = np.array([[[[1.0, 2.0, 3.0]]]]); z = np.array([[[[3.0, 4.0, 5.0]]]])
    s = np.array([[[[0.5]]]])
    kx, ks = O.build_kf_coord(x, s, z, s)
    assert np.allclose(kx, (x + z) / 2)
    assert np.allclose(ks, 0.5 / np.sqrt(2))


def test_kalman_limits():
    x = np.zeros((1, 1, 1, 3)); z = np.ones((1, 1, 1, 3))
    kx, ks = O.build_kf_coord(x, np.full((1, 1, 1, 1), 1.0), z, np.full((1, 1, 1, 1), 1e-8))
    assert np.allclose(kx, z, atol=1e-12) and ks[0, 0, 0, 0] < 1e-7
    kx, ks = O.build_kf_coord(x, np.full((1, 1, 1, 1), 1e-3), z, np.full((1, 1, 1, 1), 1e3))
    assert np.allclose(kx, x, atol=1e-9) and np.isclose(ks[0, 0, 0, 0], 1e-3, rtol=1e-9)


def test_predict_clamp():  # KFNet/KFNet.py:393-401
    h, w = 4, 5
    prob = np.full((h * w, 64), 1 / 64.0)
    offs = O.coord_volume(np.zeros((1, h, w, 2)), np.zeros((1, h, w, 2)))[1]
    tx, ts, flow = O.process_model(prob, np.zeros((h * w, 1)), offs, np.zeros((1, h, w, 3)),
                                   np.zeros((1, h, w, 1)))
    assert np.allclose(ts, np.sqrt(2e-10))
    assert np.allclose(flow, -0.5)  # App. E.7 soft-argmax bias of the -4..3 window


def test_sampler():  # tools/util.py:36-93, App. A6
    W = 8
    img = np.tile(np.arange(1, W + 1, dtype=np.float64)[None, None, :, None], (1, 3, 1, 1))
    def s(x, y=1.0):
        return O.bilinear_sampler(img, np.array([[[[x, y]]]]))[0, 0, 0, 0]
    assert s(3.0) == 4.0
    assert np.isclose(s(6.75), 7.75)
    assert s(7.0) == 0.0          # x == W-1 exactly -> weights cancel
    assert s(-0.25) == 0.0
    assert s(3.0, 2.0) == 0.0     # y == H-1
    assert s(3.0, -1.0) == 0.0
    # fp32 torch restatement agrees
    c = np.array([[[[6.75, 1.0], [7.0, 1.0], [2.5, 0.5]]]], dtype=np.float32)
    assert np.allclose(OT.bilinear_sampler(img.astype(np.float32), c), O.bilinear_sampler(img, c.astype(np.float64)))


def test_cost_volume_identity_and_direction():  # KFNet/KFNet.py:343-359
    rng = np.random.default_rng(0)
    f = rng.normal(size=(1, 6, 7, 4))
    V, offs = O.coord_volume(f, f, 8)
    V = V.reshape(6, 7, 8, 8, 4)
    assert np.all(V[:, :, 4, 4, :] == 0)
    assert offs.shape == (64, 2) and tuple(offs[0]) == (-4, -4) and tuple(offs[63]) == (3, 3)
    assert tuple(offs[1]) == (-3, -4)  # (x, y): j fastest
    # border: shifted f1 out of range -> V == f2
    assert np.all(V[0, 0, 0, 0, :] == f[0, 0, 0, :])
    # delta at (x0,y0): block (i,j) of V is -delta at (x0-(j-4), y0-(i-4))
    d = np.zeros((1, 6, 7, 1)); d[0, 3, 4, 0] = 1.0
    Vd, _ = O.coord_volume(d, np.zeros_like(d), 8)
    Vd = Vd.reshape(6, 7, 8, 8)
    i, j = 2, 7
    ys, xs = np.nonzero(Vd[:, :, i, j])
    assert list(zip(ys, xs)) == [(3 - (i - 4), 4 - (j - 4))] and Vd[5, 1, i, j] == -1.0
    Vt, _ = OT.coord_volume(f.astype(np.float32), (f * 2).astype(np.float32), 8)
    Vn, _ = O.coord_volume(f.astype(np.float32), (f * 2).astype(np.float32), 8)
    assert np.array_equal(Vt, Vn)


def test_stride2_same():  # App. E.8
    x = np.ones((1, 6, 8, 1)); w = np.ones((3, 3, 1, 1))
    y = O.conv2d_same(x, w, None, 2, False)[0, :, :, 0]
    assert y.shape == (3, 4)
    assert y[0, 0] == 9 and y[-1, 0] == 6 and y[0, -1] == 6 and y[-1, -1] == 4
    # odd size: pad (1,1)
    y = O.conv2d_same(np.ones((1, 5, 5, 1)), w, None, 2, False)[0, :, :, 0]
    assert y.shape == (3, 3) and y[0, 0] == 4 and y[1, 1] == 9
    assert O.same_pad(540, 3, 2) == (270, 0, 1) and O.same_pad(135, 3, 2) == (68, 1, 1)


def test_deconv():  # App. E.9 / A2
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, 1, 1, 5)); w = rng.normal(size=(3, 3, 4, 5)); b = rng.normal(size=4)
    y = O.conv2d_transpose_same(x, w, b, 2, False)
    ref = np.einsum('bc,ijoc->bijo', x[:, 0, 0], w[0:2, 0:2]) + b
    assert np.allclose(y, ref)
    # gradient-of-conv definition on a 2x3 input
    x = rng.normal(size=(1, 2, 3, 2)); w = rng.normal(size=(3, 3, 3, 2))
    y = O.conv2d_transpose_same(x, w, None, 2, False)
    g = rng.normal(size=y.shape)
    # <deconv(x), g> == <x, conv_s2(g, w')> where w' [kh,kw,Cin=3,Cout=2] = w
    c = O.conv2d_same(g, w, None, 2, False)
    assert np.isclose((y * g).sum(), (x * c).sum())
    yt = OT.deconv_same(OT._t(x.astype(np.float32)).permute(0, 3, 1, 2), w.astype(np.float32), None, 2, False)
    assert np.allclose(yt.permute(0, 2, 3, 1).numpy(), y, atol=1e-5)


def test_l2norm_and_transform():
    assert np.all(O.l2_normalize(np.zeros((1, 1, 1, 4))) == 0)
    v = np.array([[[[0.6, 0.8]]]])
    assert np.allclose(O.l2_normalize(v), v)
    c = np.random.default_rng(2).normal(size=(1, 2, 2, 3))
    assert np.allclose(O.apply_transform(c, np.eye(4)), c)
    T = np.eye(4); T[:3, 3] = [1, 2, 3]
    assert np.allclose(O.apply_transform(c, T), c + np.array([1, 2, 3]))


def test_conv_vs_torch_random():
    rng = np.random.default_rng(3)
    for (H, W, ci, co, k, s) in [(7, 9, 3, 5, 3, 1), (8, 10, 4, 6, 3, 2), (9, 7, 4, 2, 3, 2), (5, 5, 8, 3, 1, 1)]:
        x = rng.normal(size=(2, H, W, ci)).astype(np.float32)
        w = rng.normal(size=(k, k, ci, co)).astype(np.float32)
        b = rng.normal(size=co).astype(np.float32)
        y = O.conv2d_same(x.astype(np.float64), w, b, s, True)
        yt = OT.conv_same(OT._t(x).permute(0, 3, 1, 2), w, b, s, True).permute(0, 2, 3, 1).numpy()
        assert y.shape == yt.shape
        assert np.allclose(y, yt, atol=1e-4)


def _tiny_sequence(T=4):
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(3)
    imgs = synthetic_sequence(T, 32, 48, seed=5)
    return W, imgs, O.get_transform(synthetic_transform())


def test_reset_step_emits_the_measurement():  # App. E.12, KFNet/eval.py:94-101
    """At i % reset_period == 0 the record is concat(T.z, 1/sigma_z) whatever happened before,
    and the recursion restarts from the measurement: frames after a reset equal the frames of
    a sequence that STARTS there."""
    W, imgs, T4 = _tiny_sequence(4)
    rec, dbg = O.eval_sequence(imgs, W, T4, reset_period=2, dtype=np.float32, return_debug=True)
    for i in (0, 2):
        z, sz = dbg[i]['z'], dbg[i]['sz']
        want = np.concatenate([O.apply_transform(z, T4)[0], 1.0 / sz[0]], -1).astype(np.float32)
        assert np.array_equal(rec[i], want)
    # the reset frame's record does not depend on history; the next frame only on the pair
    # (its flow features come from frame 2, which the restarted sequence sees as frame 0)
    rec_restart = O.eval_sequence(imgs[2:], W, T4, reset_period=2, dtype=np.float32)
    assert np.array_equal(rec[2], rec_restart[0])
    assert np.array_equal(rec[3], rec_restart[1])


def test_nis_gate_touches_the_output_only():  # App. E.13, KFNet/eval.py:87-92,103-104
    from kfnet_amd.synth import synthetic_sequence
    W, imgs, T4 = _tiny_sequence(3)
    # confident measurements (sigma_z = exp(-4 + ...)) of a scene that jumps at frame 2, so
    # that the innovation test fails for many pixels
    W = dict(W)
    b = W['ScoreNet/prediction/bias'].copy()
    b[3] = -4.0
    W['ScoreNet/prediction/bias'] = b
    imgs = np.concatenate([imgs[:2], synthetic_sequence(1, 32, 48, seed=99)])
    plain, d0 = O.eval_sequence(imgs, W, T4, reset_period=500, nis_gate=False, dtype=np.float32, return_debug=True)
    gated, d1 = O.eval_sequence(imgs, W, T4, reset_period=500, nis_gate=True, dtype=np.float32, return_debug=True)
    changed = 0
    for i in (1, 2):
        # the state that is fed forward is the raw KF estimate in both runs
        assert np.array_equal(d0[i]['kf_x'], d1[i]['kf_x']) and np.array_equal(d0[i]['kf_s'], d1[i]['kf_s'])
        over = d1[i]['nis'].sum(-1)[0] > 7.815
        tz = O.apply_transform(d1[i]['z'], T4)[0].astype(np.float32)
        assert np.array_equal(gated[i][..., :3][over], tz[over])          # gated pixels show T.z
        assert np.array_equal(gated[i][..., :3][~over], plain[i][..., :3][~over])
        assert np.array_equal(gated[i][..., 3], plain[i][..., 3])         # confidence stays the KF one
        changed += int(over.sum())
    assert changed > 0, 'the synthetic sequence should trip the gate somewhere'


def test_cost_volume_factorisation_identity():
    """The algebra behind kfn_cost_volume_gather, checked on the CPU with the oracle's own
    convolution: conv0(BuildCoordVolume(f1,f2))[p,ci,cj] == b + S_k f2[p] - G_k(p+(ci-4,cj-4))
    with the class kernels of kfnet_amd.graph.cvol_class_kernels (KFNet/KFNet.py:343-359,
    cnn_wrapper/OFlowNet.py:19), including image borders and window borders."""
    from kfnet_amd.graph import cvol_class_kernels
    rng = np.random.default_rng(5)
    H, W, Cc, co = 5, 7, 8, 6
    f1 = rng.normal(size=(1, H, W, Cc))
    f2 = rng.normal(size=(1, H, W, Cc))
    wt = rng.normal(size=(3, 3, Cc, co)) / 8
    b = rng.normal(size=co)
    vol, _ = O.coord_volume(f1, f2, 8)                               # [H*W, 8, 8, Cc]
    ref = O.conv2d_same(vol, wt, b, 1, False).reshape(H, W, 8, 8, co)
    w9, s9 = cvol_class_kernels(wt)
    f1p = np.pad(f1, ((0, 0), (2, 2), (2, 2), (0, 0)))
    Gp = O.conv2d_same(f1p, w9.astype(np.float64), None, 1, False)[0]      # [H+4, W+4, 9*co]
    T = O.conv2d_same(f2, s9.astype(np.float64), np.tile(b, 9), 1, False)[0]  # [H, W, 9*co]
    cls = lambda c: 0 if c == 0 else (2 if c == 7 else 1)
    got = np.zeros_like(ref)
    for y in range(H):
        for x in range(W):
            for ci in range(8):
                for cj in range(8):
                    k = cls(ci) * 3 + cls(cj)
                    a, bb = y + ci - 2, x + cj - 2
                    g = Gp[a, bb, k * co:(k + 1) * co] if 0 <= a < H + 4 and 0 <= bb < W + 4 else 0.0
                    got[y, x, ci, cj] = T[y, x, k * co:(k + 1) * co] - g
    assert np.abs(got - ref).max() < 1e-5     # the class kernels are stored in fp32


def test_polyphase_f22_stride2_identity():
    """The algebra behind kfn_conv2d_winograd_s2 (csrc/kfn_wino_s2.hip), checked on the CPU against the oracle's
    own stride-2 convolution with the weight fragments kfnet_amd.graph.pack_winograd_s2_kernel produces: the four
    polyphase filters under F(2,2) (25 products per 2x2 outputs) accumulated into NINE accumulators
    (D00 D01 D10 D11 | R0 R1 | C0 C1 | Z) and Y00 = D00+R0+C0+Z, Y01 = D01+R0+C1+Z, Y10 = D10+R1+C0+Z,
    Y11 = D11+R1+C1+Z.  Slot / fragment / accumulator tables as in the kernel."""
    from kfnet_amd.graph import pack_winograd_s2_kernel
    rng = np.random.default_rng(11)
    n, H, W, ci, co = 2, 8, 12, 8, 5
    x = rng.normal(size=(n, H, W, ci))
    wt = rng.normal(size=(3, 3, ci, co)).astype(np.float32)
    b = rng.normal(size=co)
    ref = O.conv2d_same(x, wt.astype(np.float64), b, 2, False)
    u = pack_winograd_s2_kernel(wt)                                  # [ci/8][16][cout_pad][8]
    U = u.transpose(1, 0, 3, 2).reshape(16, ci, u.shape[2])[:, :, :co].astype(np.float64)   # [frag][ci][co]
    D00, D01, D10, D11, R0, R1, C0, C1, Z = range(9)
    POS_SLOT = [0, 2, 6, 8, 20, 13, 14, 17, 18, 23, 19, 21, 22, 24, 15, 9, 10, 11, 12, 16, 4, 1, 7, 3, 5]
    POS_FRAG = [0, 2, 6, 8, 13, 9, 9, 11, 11, 13, 12, 14, 12, 14, 10, 15, 15, 15, 15, 10, 4, 1, 7, 3, 5]
    POS_ACC = [D00, D01, D10, D11, R0, D00, D01, D10, D11, R1, D00, D01, D10, D11, C0, D00, D01, D10, D11, C1,
               Z, R0, R1, C0, C1]
    bt = lambda d0, d1, d2: (d0 - d1, d1, d1 - d2)
    xp = np.pad(x, ((0, 0), (0, 4), (0, 4), (0, 0)))                 # zeros after the image
    got = np.zeros_like(ref)
    for im in range(n):
        for ty in range((H // 2 + 1) // 2):
            for tx in range((W // 2 + 1) // 2):
                d = xp[im, 4 * ty:4 * ty + 5, 4 * tx:4 * tx + 5]     # 5x5 patch
                slot = [None] * 25
                e = [[d[2 * m, 2 * k] for k in range(3)] for m in range(3)]          # (even,even): B^T e B
                e = [list(bt(*row)) for row in e]
                cols = [bt(e[0][k], e[1][k], e[2][k]) for k in range(3)]
                for m in range(3):
                    for k in range(3):
                        slot[3 * m + k] = cols[k][m]
                for m in range(2):                                                    # (odd,odd)
                    for k in range(2):
                        slot[9 + 2 * m + k] = d[2 * m + 1, 2 * k + 1]
                for k in range(2):                                                    # (even,odd): along rows
                    t = bt(d[0, 2 * k + 1], d[2, 2 * k + 1], d[4, 2 * k + 1])
                    for m in range(3):
                        slot[13 + 2 * m + k] = t[m]
                for m in range(2):                                                    # (odd,even): along columns
                    t = bt(d[2 * m + 1, 0], d[2 * m + 1, 2], d[2 * m + 1, 4])
                    for k in range(3):
                        slot[19 + 3 * m + k] = t[k]
                acc = [np.zeros(co) for _ in range(9)]
                for a in (D00, D01, D10, D11):
                    acc[a] += b
                for p in range(25):
                    acc[POS_ACC[p]] += slot[POS_SLOT[p]] @ U[POS_FRAG[p]]
                y = [[acc[D00] + acc[R0] + acc[C0] + acc[Z], acc[D01] + acc[R0] + acc[C1] + acc[Z]],
                     [acc[D10] + acc[R1] + acc[C0] + acc[Z], acc[D11] + acc[R1] + acc[C1] + acc[Z]]]
                for di in range(2):
                    for dj in range(2):
                        oy, ox = 2 * ty + di, 2 * tx + dj
                        if oy < H // 2 and ox < W // 2:
                            got[im, oy, ox] = y[di][dj]
    assert np.abs(got - ref).max() < 1e-5     # the fragments are stored in fp32


def test_polyphase_f42_stride2_identity():
    """The algebra behind the F(4,2) form of kfn_conv2d_winograd_s2 (csrc/kfn_wino_s2c.hip), on the CPU against the oracle's own
    stride-2 convolution with the fragments kfnet_amd.graph.pack_winograd_s2_kernel_c produces: the four polyphase filters on
    4x4 output tiles under F(4,2) -- B^T along two-tap axes, C (4 values -> accumulator indices {0,1,2,4}) along one-tap axes --
    81 products into the 25 accumulators M[xi][nu], output Y = A^T M A, bias in M[1][1].  Positions / fragments / accumulators /
    LDS order as in the kernel's tables."""
    from kfnet_amd.graph import pack_winograd_s2_kernel_c
    rng = np.random.default_rng(12)
    AT = np.array([[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 0], [0, 1, -1, 8, 1]], np.float64)
    BT = np.array([[2, -1, -2, 1, 0], [0, 2, 1, -1, 0], [0, -2, 3, -1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]], np.float64)
    CT = np.array([[1, 0, -1, 0], [0, .5, .5, 0], [0, -.5, .5, 0], [0, -1, 0, 1]], np.float64)
    IDX = [0, 1, 2, 4]
    assert np.allclose(AT[:, IDX] @ CT, np.eye(4))          # C is the inverse of A^T restricted to the indices {0,1,2,4}
    # the kernel's tables, rebuilt from their definition: round r = EE_r (5) EO_r (4) OE_r (4) OO_r (4, r < 4)
    pos = []        # (slot, weight register, accumulator, fragment)
    for r in range(5):
        pos += [(r * 5 + nu, nu, r * 5 + nu, r * 5 + nu) for nu in range(5)]
        pos += [(25 + r * 4 + j, 5, r * 5 + IDX[j], 25 + r) for j in range(4)]
        pos += [(45 + i * 5 + r, 6, IDX[i] * 5 + r, 30 + r) for i in range(4)]
        if r < 4:
            pos += [(65 + r * 4 + j, 7, IDX[r] * 5 + IDX[j], 35) for j in range(4)]
    assert len(pos) == 81 and sorted(p[0] for p in pos) == list(range(81))
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'kfnet_amd', 'csrc', 'kfn_wino_s2c.hip')).read()
    def table(name):
        m = re.search(r'constexpr int %s\[\w+\] = \{([^}]*)\}' % name, src)
        return [int(v) for v in m.group(1).replace('\n', ' ').split(',')]
    assert table('POS_BREG') == [p[1] for p in pos] and table('POS_ACC') == [p[2] for p in pos] and table('POS_FRAG') == [p[3] for p in pos]
    rank = {p[0]: i for i, p in enumerate(pos)}
    assert table('SLOT_IDX') == [rank[s] for s in range(81)]
    for (n, H, W, ci, co) in [(2, 16, 24, 16, 5), (1, 8, 8, 32, 3)]:
        x = rng.normal(size=(n, H, W, ci))
        wt = rng.normal(size=(3, 3, ci, co)).astype(np.float32)
        b = rng.normal(size=co)
        ref = O.conv2d_same(x, wt.astype(np.float64), b, 2, False)
        u = pack_winograd_s2_kernel_c(wt)                                  # [ci/16][36][cout_pad][16]
        assert u.shape == (ci // 16, 36, 32, 16) and u.dtype == np.float32
        U = u.transpose(1, 0, 3, 2).reshape(36, ci, u.shape[2])[:, :, :co].astype(np.float64)   # [frag][ci][co]
        xp = np.pad(x, ((0, 0), (0, 1), (0, 1), (0, 0)))                   # zeros after the image
        got = np.zeros_like(ref)
        for im in range(n):
            for ty in range(H // 8):
                for tx in range(W // 8):
                    d = xp[im, 8 * ty:8 * ty + 9, 8 * tx:8 * tx + 9]       # 9x9 patch
                    slot = [None] * 81
                    V = np.einsum('xm,mnc,yn->xyc', BT, d[0::2, 0::2], BT)
                    for xi in range(5):
                        for nu in range(5):
                            slot[xi * 5 + nu] = V[xi, nu]
                    V = np.einsum('xm,mjc,vj->xvc', BT, d[0::2, 1::2], CT)
                    for xi in range(5):
                        for j in range(4):
                            slot[25 + xi * 4 + j] = V[xi, j]
                    V = np.einsum('ui,inc,yn->uyc', CT, d[1::2, 0::2], BT)
                    for i in range(4):
                        for nu in range(5):
                            slot[45 + i * 5 + nu] = V[i, nu]
                    V = np.einsum('ui,ijc,vj->uvc', CT, d[1::2, 1::2], CT)
                    for i in range(4):
                        for j in range(4):
                            slot[65 + i * 4 + j] = V[i, j]
                    acc = [np.zeros(co) for _ in range(25)]
                    acc[6] += b
                    for (s, _, a, f) in pos:
                        acc[a] += slot[s] @ U[f]
                    M = np.stack(acc).reshape(5, 5, co)
                    got[im, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = np.einsum('ix,xyo,jy->ijo', AT, M, AT)
        assert np.abs(got - ref).max() < 2e-5     # the fragments are stored in fp32


def test_window_fc_matrix_identity():
    """pack_window_fc_kernel: a 3x3 SAME stride-1 convolution on 2x2 images equals one dense [4 Cin] x [4 Cout]
    matrix per window (kfnet_amd.graph.WindowFcConvOp, OFlowNet's 2x2 level)."""
    from kfnet_amd.graph import pack_bias_x4, pack_window_fc_kernel
    rng = np.random.default_rng(21)
    P, ci, co = 7, 8, 16
    x = rng.normal(size=(P, 2, 2, ci))
    wt = rng.normal(size=(3, 3, ci, co)).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    ref = O.conv2d_same(x, wt.astype(np.float64), b, 1, False)          # [P,2,2,co]
    m = pack_window_fc_kernel(wt)[:4 * co].astype(np.float64)            # [(q,o)][(p,c)]
    got = x.reshape(P, 4 * ci) @ m.T + pack_bias_x4(b)[:4 * co]
    assert np.abs(got.reshape(P, 2, 2, co) - ref).max() < 1e-5


def test_get_kf_coord2_symmetric_posterior():
    """KFNet.GetKFCoord2 (KFNet/KFNet.py:487-502): in exact arithmetic (1-K)^2 P + K^2 R == (1-K) P == P R / (P + R);
    the restatement must agree with that closed form and with build_kf_coord's mean."""
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 5, 7, 3)); z = rng.normal(size=(1, 5, 7, 3))
    s = np.abs(rng.normal(size=(1, 5, 7, 1))) + 0.1; sz = np.abs(rng.normal(size=(1, 5, 7, 1))) + 0.1
    c2, u2 = O.get_kf_coord2(x, s, z, sz)
    c1, u1 = O.build_kf_coord(x, s, z, sz)
    assert np.allclose(c2, c1, rtol=0, atol=0)
    assert np.allclose(u2 ** 2, (s ** 2) * (sz ** 2) / (s ** 2 + sz ** 2), rtol=1e-12)
    assert np.allclose(u2, u1, rtol=1e-12)
    half = O.get_kf_coord2(x, s, z, s)
    assert np.allclose(half[0], (x + z) / 2) and np.allclose(half[1], s / np.sqrt(2))


def test_winograd_f43_matrices_and_partial_output_transform():
    """The F(4x4,3x3) triple of kfnet_amd.graph (== csrc/kfn_wino4.hip) reproduces the oracle's 3x3 SAME convolution, the
    packer's layout is the documented one, and the kernel's split of the output transform -- a full nu pass, then the xi
    pass in two halves {0,1,2} / {3,4,5} whose partial 4x4 outputs are added -- equals A^T M A."""
    from kfnet_amd.graph import _WINO4_AT as AT, _WINO4_BT as BT, _WINO4_G as G, pack_winograd_f43_kernel
    rng = np.random.default_rng(43)
    x = rng.normal(size=(1, 8, 12, 8))
    w = rng.normal(size=(3, 3, 8, 5))
    ref = O.conv2d_same(x, w, None, 1, False)
    xp = np.zeros((1, 8 + 2, 12 + 2, 8))
    xp[:, 1:9, 1:13] = x
    U = np.einsum('ai,ijco,bj->abco', G, w, G)
    got = np.zeros_like(ref)
    for ty in range(2):
        for tx in range(3):
            d = xp[0, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6, :]                      # [6,6,ci]
            V = np.einsum('ar,rcx,bc->abx', BT, d, BT)
            M = np.einsum('abx,abxo->abo', V, U)                                        # [6,6,co]
            # the kernel's order: nu pass for every xi, then the two xi halves
            R = np.einsum('jn,ano->ajo', AT, M)                                         # [xi][j][co]
            P0 = np.einsum('ia,ajo->ijo', AT[:, 0:3], R[0:3])
            P1 = np.einsum('ia,ajo->ijo', AT[:, 3:6], R[3:6])
            got[0, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4, :] = P0 + P1
    assert np.abs(got - ref).max() < 1e-10
    wf = rng.normal(size=(3, 3, 16, 40)).astype(np.float32)
    u4 = pack_winograd_f43_kernel(wf)
    assert u4.shape == (2, 36, 64, 8) and u4.dtype == np.float32
    Uf = np.einsum('ai,ijco,bj->abco', G, wf.astype(np.float64), G)
    for (ci, co, xi, nu) in ((0, 0, 0, 0), (9, 39, 3, 5), (15, 17, 5, 1)):
        assert u4[ci // 8, 6 * xi + nu, co, ci % 8] == np.float32(Uf[xi, nu, ci, co])
    assert np.all(u4[:, :, 40:, :] == 0)
