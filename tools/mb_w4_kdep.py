"""Cost per workgroup of the F(4x4,3x3) kernel: one layer shape timed at Cin = 64 .. 512 (same tiles, same Cout) -- the slope is the
super-step, the intercept what a workgroup costs beyond its super-steps (launch, prologue, epilogue, output stores).
    MB_SHAPE=480,640,64 MB_BATCH=20 python tools/mb_w4_kdep.py      (H, W, Cout)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kfnet_amd import _lib
if os.environ.get('MB_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MB_LIB'])
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
N = int(os.environ.get('MB_BATCH', '20'))
H, W, co = [int(v) for v in os.environ.get('MB_SHAPE', '480,640,64').split(',')]
res = []
for ci in (64, 128, 256, 512):
    x = torch.randn(N * H * W * ci, device='cuda')
    u = torch.randn(36 * co * ci, device='cuda') * 0.02
    y = torch.empty(N * H * W * co, device='cuda')
    d = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=-(-co // 32) * 32, ldy=co, kh=3, kw=3, stride=1, relu=1,
                      wino_form=int(os.environ.get('MB_F43_FORM', '3')))
    try:
        t = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd_f43(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), st), 'w4'))
    except _lib.KfnError as e:          # (two images of the input beyond the 32-bit offsets of the kernel)
        print('Cin %4d: %s' % (ci, str(e)[:60]))
        break
    wgs = N * (-(-(H // 4) // 8)) * 1  # informative only
    print('Cin %4d: %.3f ms' % (ci, t), flush=True)
    res.append((ci, t))
    del x, u, y
(c0, t0), (c1, t1) = res[1], res[-1]
slope = (t1 - t0) / ((c1 - c0) / 16)
print('per super-step of 16 channels: %.4f ms; intercept at Cin -> 0: %.3f ms (%.0f %% of the Cin = 64 launch)'
      % (slope, res[0][1] - slope * res[0][0] / 16, 100 * (res[0][1] - slope * res[0][0] / 16) / res[0][1]))
