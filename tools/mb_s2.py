"""Microbenchmark of SCoordNet's 3x3 stride-2 layers at the bench batch: the polyphase F(2,2) kernel
(kfn_conv2d_winograd_s2, 25/36 of the nominal MFMAs) vs the direct implicit GEMM."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kfnet_amd import _lib
if os.environ.get('MB_LIB'):          # an A/B build (tools/mb/build_hot.sh)
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
ORDER = int(os.environ.get('MB_WINO_ORDER', '0'))   # kfn_conv_desc.wino_order: 0 default, 1 tile blocks fastest, 2 channel groups fastest
N = int(os.environ.get('MB_BATCH', '16'))
LAYERS = [('conv2a', 480, 640, 64, 256), ('conv3a', 240, 320, 256, 512), ('conv4a', 120, 160, 512, 1024)]
if os.environ.get('MB_LAYERS'):      # e.g. MB_LAYERS='k64,480,640,64,256;k128,480,640,128,256': fixed cost per workgroup = the intercept over Cin
    LAYERS = [(f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4])) for f in (t.split(',') for t in os.environ['MB_LAYERS'].split(';'))]
for (name, H, W, ci, co) in LAYERS:
    x = torch.randn(N * H * W * ci, device='cuda')
    FORM = int(os.environ.get('MB_S2_FORM', '0'))       # 4 = the eight-wave form (wino_s2b_kernel), 5 = polyphase + F(4,2) (wino_s2c_kernel)
    u = torch.randn((36 if FORM == 5 else 16) * co * ci, device='cuda') * 0.02
    w9 = torch.randn(co * 9 * ci, device='cuda') * 0.02
    y = torch.empty(N * (H // 2) * (W // 2) * co, device='cuda')
    F16 = os.environ.get('MB_F16', '') == '1'
    d = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=3, kw=3, stride=2, relu=1, wino_order=ORDER,
                      wino_form=FORM)
    d16 = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=3, kw=3, stride=2, relu=1, operand_dtype=_lib.OPERAND_F16, wino_order=ORDER)
    if F16:
        u = u.half(); w9 = w9.half()
    lay = os.environ.get('MB_LAYOUT', '00')       # 'xy' digits, 1 = KFN_LAYOUT_C16 (form 5 only; timing -- the buffers hold noise either way)
    d.x_layout, d.y_layout = int(lay[0]), int(lay[1])
    t_s2 = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd_s2(C.byref(d16 if F16 else d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), st), 's2'))
    d.x_layout, d.y_layout = 0, 0
    t_dir = timeit(lambda: _lib.check(lib.kfn_conv2d_nhwc(C.byref(d16 if F16 else d), x.data_ptr(), w9.data_ptr(), None, y.data_ptr(), st), 'c'))
    nominal = 2.0 * N * (H // 2) * (W // 2) * 9 * ci * co
    print('%-7s %3dx%3d C%4d->%4d: polyphase %.3f ms (%.1f TF executed, %.1f nominal) | direct %.3f ms (%.1f TF)'
          % (name, H, W, ci, co, t_s2, nominal * (81 / 144 if FORM == 5 else 25 / 36) / t_s2 / 1e9, nominal / t_s2 / 1e9, t_dir, nominal / t_dir / 1e9), flush=True)
    del x, u, y, w9
