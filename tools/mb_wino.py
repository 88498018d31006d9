"""Microbenchmark: Winograd GEMM phase vs an equal-MFMA-work 1x1 conv (materialised-V ceiling)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kfnet_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for (N, H, W, ci, co) in [(17, 60, 80, 1024, 1024), (17, 120, 160, 512, 512), (17, 240, 320, 256, 256)]:
    x = torch.randn(N * H * W * ci, device='cuda')
    u = torch.randn(16 * co * ci, device='cuda') * 0.02
    y = torch.empty(N * H * W * co, device='cuda')
    Mt = N * (H // 2) * (W // 2)
    ws = torch.empty(16 * Mt * co, device='cuda')
    d = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=3, kw=3, stride=1, relu=1,
                      config=int(os.environ.get('KFN_WINO_CFG', '0')))
    t_gemm = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), ws.data_ptr(), 1, st), 'w'))
    t_out = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), ws.data_ptr(), 2, st), 'w'))
    t_fused = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd_fused(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), st), 'wf'))
    fl = 2.0 * 16 * Mt * ci * co
    # 1x1 conv with the same MFMA work: M = 16*Mt rows
    v = torch.randn(16 * Mt * ci, device='cuda')
    d1 = _lib.ConvDesc(N=16, H=Mt // 16 if Mt % 16 == 0 else 1, W=16, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=1, kw=1, stride=1, relu=0)
    d1.H = Mt // 16
    t_1x1 = timeit(lambda: _lib.check(lib.kfn_conv2d_nhwc(C.byref(d1), v.data_ptr(), u.data_ptr(), None, ws.data_ptr(), st), 'c'))
    dd = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=3, kw=3, stride=1, relu=1)
    w9 = torch.randn(co * 9 * ci, device='cuda') * 0.02
    t_dir = timeit(lambda: _lib.check(lib.kfn_conv2d_nhwc(C.byref(dd), x.data_ptr(), w9.data_ptr(), None, y.data_ptr(), st), 'c'))
    print('%dx%d C%d: FUSED %.3f ms (%.1f TF exec) | wino gemm %.3f ms (%.1f TF exec), out %.3f ms | 1x1 same-work %.3f ms (%.1f TF) | direct %.3f ms (%.1f TF)'
          % (H, W, ci, t_fused, fl / t_fused / 1e9, t_gemm, fl / t_gemm / 1e9, t_out, t_1x1, fl / t_1x1 / 1e9, t_dir, fl * 2.25 / t_dir / 1e9))
    del x, u, y, ws, v, w9
