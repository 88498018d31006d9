"""Does the Winograd workspace round trip get cheaper when it fits the 256 MB Infinity Cache?
Times GEMM phase and output transform back to back for conv2b-shaped problems of growing size."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kfnet_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for (N, H, W, ci, co) in [(1, 32, 320, 256, 256), (1, 64, 320, 256, 256), (1, 128, 320, 256, 256), (1, 240, 320, 256, 256),
                          (2, 240, 320, 256, 256), (4, 240, 320, 256, 256), (17, 240, 320, 256, 256)]:
    x = torch.randn(N * H * W * ci, device='cuda')
    u = torch.randn(16 * co * ci, device='cuda') * 0.02
    y = torch.empty(N * H * W * co, device='cuda')
    Mt = N * (H // 2) * (W // 2)
    ws = torch.empty(16 * Mt * co, device='cuda')
    d = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=3, kw=3, stride=1, relu=1)
    both = lambda: _lib.check(lib.kfn_conv2d_winograd(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), ws.data_ptr(), 3, st), 'w')
    g = lambda: _lib.check(lib.kfn_conv2d_winograd(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), ws.data_ptr(), 1, st), 'w')
    o = lambda: _lib.check(lib.kfn_conv2d_winograd(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), ws.data_ptr(), 2, st), 'w')
    tb, tg, to = timeit(both), timeit(g), timeit(o)
    wsmb = 16 * Mt * co * 4 / 1e6
    print('ws %7.1f MB (Mt=%7d): gemm+out %.3f ms | gemm alone %.3f | out alone %.3f (%.2f TB/s) | per-Mtile: both %.2f ns gemm %.2f out %.2f'
          % (wsmb, Mt, tb, tg, to, (wsmb * 1e6 + N * H * W * co * 4) / to / 1e9, tb / Mt * 1e6, tg / Mt * 1e6, to / Mt * 1e6))
    del x, u, y, ws
