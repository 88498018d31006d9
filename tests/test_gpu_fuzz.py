"""-m gpu randomized shape sweep of the MFMA conv entry points against PyTorch's own GPU
convolution (MIOpen/rocBLAS, fp32) used purely as a second, independent reference for
many shapes the fp64 numpy oracle would be slow on: tile-edge rows/columns, odd sizes,
both strides, all three operand modes, Winograd."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch_same_conv(x, w, b, stride, relu):
    import torch
    import torch.nn.functional as F
    n, h, wd, ci = x.shape
    k = w.shape[0]
    def pad(n_, s):
        out = -(-n_ // s)
        tot = max((out - 1) * s + k - n_, 0)
        return tot // 2, tot - tot // 2
    pt, pb = pad(h, stride)
    pl, pr = pad(wd, stride)
    xt = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xt.double(), w.permute(3, 2, 0, 1).double(), b.double(), stride=stride)
    y = y.permute(0, 2, 3, 1)
    return (torch.relu(y) if relu else y).contiguous()


def _shapes(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        ci = int(rng.choice([16, 32, 48, 64, 96, 128, 160, 256]))
        co = int(rng.choice([1, 4, 16, 20, 32, 33, 64, 100, 128, 192, 256, 300]))
        k = int(rng.choice([1, 3, 3, 3]))
        s = int(rng.choice([1, 1, 2]))
        n = int(rng.integers(1, 5))
        h = int(rng.integers(1, 40))
        w = int(rng.integers(1, 40))
        out.append((n, h, w, ci, co, k, s, bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_conv_random_shapes_all_modes(seed):
    import torch
    from kfnet_amd import _lib
    from kfnet_amd.graph import pack_conv_kernel, pack_winograd_kernel
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cuda').manual_seed(seed)
    for (n, h, w, ci, co, k, s, relu) in _shapes(seed, 40):
        x = torch.relu(torch.randn(n, h, w, ci, device='cuda', generator=g)) * 2
        wt = torch.randn(k, k, ci, co, device='cuda', generator=g) * float(np.sqrt(2.0 / (k * k * ci)))
        b = torch.randn(co, device='cuda', generator=g)
        ref = _torch_same_conv(x, wt, b, s, relu)
        scale = float(ref.abs().max()) + 1.0
        ho, wo = ref.shape[1], ref.shape[2]
        wp = pack_conv_kernel(wt.cpu().numpy())
        cp = wp.shape[0]
        modes = [(0, torch.from_numpy(wp).cuda(), 3e-6)]
        if ci % 32 == 0:
            m = wp * np.float32(1024.0)
            hi = m.astype(np.float16)
            x3 = np.stack([hi, (m - hi.astype(np.float32)).astype(np.float16)])
            modes.append((2, torch.from_numpy(x3).cuda(), 1e-5))
            modes.append((1, torch.from_numpy(wp.astype(np.float16)).cuda(), 6e-3))
        for od, wdev, rtol in modes:
            y = torch.empty(n * ho * wo * co, device='cuda')
            d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=cp, ldy=co, kh=k, kw=k, stride=s,
                              relu=int(relu), operand_dtype=od)
            _lib.check(lib.kfn_conv2d_nhwc(C.byref(d), x.data_ptr(), wdev.data_ptr(), b.data_ptr(), y.data_ptr(), st),
                       'conv %r mode %d' % ((n, h, w, ci, co, k, s), od))
            err = float((y.view_as(ref).double() - ref).abs().max())
            assert err <= rtol * scale, ('direct', (n, h, w, ci, co, k, s, relu), od, err, scale)
        if k == 3 and s == 1 and co % 4 == 0:
            u = torch.from_numpy(pack_winograd_kernel(wt.cpu().numpy())).cuda()
            d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ci, Cout=co, cout_pad=cp, ldy=co, kh=3, kw=3, stride=1,
                              relu=int(relu))
            nb = C.c_size_t()
            _lib.check(lib.kfn_winograd_workspace_bytes(C.byref(d), C.byref(nb)), 'ws')
            ws = torch.empty(max(nb.value // 4, 4), device='cuda')
            y = torch.empty(n * ho * wo * co, device='cuda')
            _lib.check(lib.kfn_conv2d_winograd(C.byref(d), x.data_ptr(), u.data_ptr(), b.data_ptr(), y.data_ptr(),
                                               ws.data_ptr(), 3, st), 'wino')
            err = float((y.view_as(ref).double() - ref).abs().max())
            assert err <= 6e-6 * scale, ('wino', (n, h, w, ci, co, relu), err, scale)
    torch.cuda.synchronize()
