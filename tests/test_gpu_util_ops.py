"""The reference's graph-level helpers as importable callables: kfnet_amd.KFNet.util.ApplyTransform /
GetPixelMap (KFNet/util.py:12-63) and kfnet_amd.tools.util.bilinear_sampler (tools/util.py:3-94) against the oracle's
restatements, through the Python-level API (graph ops) and straight through the C ABI."""
import numpy as np
import pytest

from oracle import kfnet_oracle as O

pytestmark = pytest.mark.gpu


def _graph():
    from kfnet_amd.graph import Graph
    return Graph()


def _run(g):
    import torch
    g.finalize('cuda:0')
    g.run(torch.cuda.current_stream().cuda_stream, g.ops)
    torch.cuda.synchronize()


def test_apply_transform_vs_oracle_whole_tensor_view_and_batched():
    """4x4 for the whole batch, a channel VIEW of a packed [B,h,w,4] buffer as input (how eval.py's maps are stored), a
    per-image Bx4x4, and inverse=True; tolerance: fp32 rounding of three products and three sums (1e-6 relative)."""
    from kfnet_amd.KFNet.util import ApplyTransform
    from kfnet_amd.synth import synthetic_transform
    rng = np.random.default_rng(5)
    B, h, w = 3, 7, 9
    packed = rng.normal(size=(B, h, w, 4)).astype(np.float32) * 3.0
    T = synthetic_transform().astype(np.float32)
    Tb = np.stack([T, np.linalg.inv(T).astype(np.float32), np.eye(4, dtype=np.float32)])
    g = _graph()
    src = g.placeholder((B, h, w, 4), name='packed')
    coords = src.channels(0, 3)
    y1 = ApplyTransform(coords, T)
    y2 = ApplyTransform(coords, Tb)
    y3 = ApplyTransform(coords, T, inverse=True)
    assert y1.get_shape().as_list() == [B, h, w, 3]
    g.finalize('cuda:0')
    src.upload(packed)
    _run(g)
    x64 = packed[..., :3].astype(np.float64)
    ref1 = O.apply_transform(x64, T.astype(np.float64))
    scale = np.abs(ref1).max()
    assert np.abs(y1.numpy() - ref1).max() <= 2e-6 * scale
    for b in range(B):
        refb = O.apply_transform(x64[b], Tb[b].astype(np.float64))
        assert np.abs(y2.numpy()[b] - refb).max() <= 2e-6 * max(scale, np.abs(refb).max())
    ref3 = O.apply_transform(x64, np.linalg.inv(T).astype(np.float32).astype(np.float64))
    assert np.abs(y3.numpy() - ref3).max() <= 2e-6 * np.abs(ref3).max()
    with pytest.raises(ValueError):
        ApplyTransform(src, T)                       # BxHxWx4 is not a coordinate map
    with pytest.raises(ValueError):
        ApplyTransform(coords, np.eye(3))


def test_get_pixel_map_is_exact():
    from kfnet_amd.KFNet.KFNet import KFNetDataSpec
    from kfnet_amd.KFNet.util import GetPixelMap
    g = _graph()
    m = GetPixelMap(2, 5, 8, graph=g)
    spec = KFNetDataSpec()
    mn = GetPixelMap(1, 6, 4, normalize=True, spec=spec, graph=g)
    _run(g)
    ref = O.get_pixel_map(5, 8, np.float32)
    got = m.numpy()
    assert got.shape == (2, 5, 8, 2) and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[0])
    assert got[0, 3, 6, 0] == 6.0 and got[0, 3, 6, 1] == 3.0           # (x, y) = (column, row)
    r = O.get_pixel_map(6, 4, np.float32)[0]
    want = np.stack([(r[..., 0] - np.float32(spec.u)) / np.float32(spec.focal_x),
                     (r[..., 1] - np.float32(spec.v)) / np.float32(spec.focal_y)], axis=-1)
    assert np.array_equal(mn.numpy()[0], want)
    with pytest.raises(ValueError):
        GetPixelMap(1, 2, 2)                         # no implicit default graph


@pytest.mark.parametrize('shape', [(1, 60, 80, 4, 60, 80), (2, 9, 8, 3, 5, 7), (1, 8, 8, 1, 8, 8)])
def test_bilinear_sampler_vs_oracle_with_the_border_rule(shape):
    """Interior samples, samples left / right / above / below the image, exact integer positions and THE border rule of
    tools/util.py:55-63: x = Ws-1 (and anything beyond) evaluates to 0 because the weights come from the clamped corners."""
    from kfnet_amd.tools.util import bilinear_sampler
    B, Hs, Ws, C, Ht, Wt = shape
    rng = np.random.default_rng(B * 1000 + Hs)
    imgs = rng.normal(size=(B, Hs, Ws, C)).astype(np.float32)
    co = np.stack([rng.uniform(-2.0, Ws + 1.0, size=(B, Ht, Wt)), rng.uniform(-2.0, Hs + 1.0, size=(B, Ht, Wt))], axis=-1).astype(np.float32)
    co[:, 0, 0] = (Ws - 1.0, 1.25)          # x = W-1 -> 0
    co[:, 0, 1] = (Ws - 1.25, 1.5)          # just inside -> ordinary bilinear
    co[:, 0, 2] = (2.0, 3.0)                # exact pixel
    co[:, 1, 0] = (1.5, Hs - 1.0)           # y = H-1 -> 0
    co[:, 1, 1] = (-0.25, 2.0)              # x < 0 -> 0
    g = _graph()
    ti = g.placeholder((B, Hs, Ws, C), name='imgs')
    tc = g.placeholder((B, Ht, Wt, 2), name='coords')
    y = bilinear_sampler(ti, tc)
    g.finalize('cuda:0')
    ti.upload(imgs)
    tc.upload(co)
    _run(g)
    got = y.numpy()
    for b in range(B):       # the oracle's sampler is written for one image (eval.py's state)
        ref = O.bilinear_sampler(imgs[b:b + 1], co[b:b + 1])[0]
        assert np.abs(got[b] - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), b
        ref64 = O.bilinear_sampler(imgs[b:b + 1].astype(np.float64), co[b:b + 1].astype(np.float64))[0]
        assert np.abs(got[b] - ref64).max() <= 2e-5
    assert np.all(got[:, 0, 0] == 0.0) and np.all(got[:, 1, 0] == 0.0) and np.all(got[:, 1, 1] == 0.0)
    assert np.any(got[:, 0, 1] != 0.0)
    assert np.allclose(got[:, 0, 2], imgs[:, 3, 2], rtol=0, atol=1e-6)


def test_helpers_reproduce_the_scan_kernels_process_model():
    """BuildOFlowNet's tail rebuilt from the Python-level helpers -- pixel_map = GetPixelMap + flow is done on the host here,
    then bilinear_sampler(last_coord), bilinear_sampler(last_uncertainty) (KFNet/KFNet.py:386-392) -- equals the `temp`
    output of the fused scan (kfn_kalman_scan_ex) bit for bit: same arithmetic, same order."""
    import ctypes as C
    import torch
    from kfnet_amd import _lib
    from kfnet_amd.KFNet.util import GetPixelMap
    from kfnet_amd.tools.util import bilinear_sampler
    from tests.gpu_util import dev, stream, sync
    lib = _lib.load()
    H, W = 12, 16
    rng = np.random.default_rng(3)
    flow = (rng.normal(size=(1, H, W, 2)) * 2.0).astype(np.float32)
    state = rng.normal(size=(1, H, W, 4)).astype(np.float32)
    state[..., 3] = np.abs(state[..., 3]) * 0.3 + 0.05
    g = _graph()
    pm = GetPixelMap(1, H, W, graph=g)
    g.finalize('cuda:0')
    g.run(torch.cuda.current_stream().cuda_stream, g.ops)
    coords = pm.numpy() + flow                                  # fp32 add, as tf.add
    g2 = _graph()
    st = g2.placeholder((1, H, W, 4), name='state')
    tc = g2.placeholder((1, H, W, 2), name='pixel_map')
    warped_x = bilinear_sampler(st.channels(0, 3), tc)
    warped_s = bilinear_sampler(st.channels(3, 1), tc)
    g2.finalize('cuda:0')
    st.upload(state)
    tc.upload(coords)
    g2.run(torch.cuda.current_stream().cuda_stream, g2.ops)
    torch.cuda.synchronize()
    # the fused kernel's prediction for the same inputs
    sig = np.full((1, H, W, 1), 0.01, np.float32)
    meas = rng.normal(size=(1, H, W, 4)).astype(np.float32)
    meas[..., 3] = 0.2
    d = _lib.KalmanDesc(S=1, T=1, H=H, W=W, t0=1, reset_period=500, min_uncertainty=1e-5, nis_gate=0.0, has_transform=0)
    fd, sd, md, std = dev(flow), dev(sig), dev(meas), dev(state.copy())
    rec = torch.zeros(H * W * 4, device='cuda')
    temp = torch.zeros(H * W * 4, device='cuda')
    need = C.c_size_t(0)
    _lib.check(lib.kfn_kalman_scan_scratch_bytes(C.byref(d), C.byref(need)), 'scratch')
    scratch = torch.empty(max(1, (need.value + 3) // 4), device='cuda')
    _lib.check(lib.kfn_kalman_scan_ex(C.byref(d), fd.data_ptr(), sd.data_ptr(), md.data_ptr(), std.data_ptr(), rec.data_ptr(),
                                      temp.data_ptr(), None, None, 0, scratch.data_ptr(), stream()), 'scan')
    sync()
    t = temp.cpu().numpy().reshape(H, W, 4)
    assert np.array_equal(t[..., :3], warped_x.numpy()[0])
    lv = np.maximum(warped_s.numpy()[0, ..., 0] ** 2, np.float32(1e-10))
    tv = np.maximum(sig[0, ..., 0] ** 2, np.float32(1e-10))
    assert np.allclose(t[..., 3], np.sqrt(tv + lv), rtol=2e-7, atol=0)


def test_c_abi_argument_checks():
    import torch
    from kfnet_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(64, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    assert lib.kfn_apply_transform(buf.data_ptr(), 2, buf.data_ptr(), 0, 1, 2, 2, buf.data_ptr(), 3, s) == -1
    assert lib.kfn_pixel_map(buf.data_ptr(), 2, 1, 2, 2, 1, 0.0, 0.0, 0.0, 1.0, s) == -1
    assert lib.kfn_bilinear_sampler(buf.data_ptr(), 1, 1, 2, 2, 2, buf.data_ptr(), 2, 2, 2, buf.data_ptr(), 2, s) == -1
    assert b'kfn_bilinear_sampler' in lib.kfn_last_error()
