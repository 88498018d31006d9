"""Microbenchmark of the 3x3 stride-1 layers of SCoordNet at the bench batch (17 frames):
single-kernel Winograd (kfn_conv2d_winograd_fused) vs the two-kernel form (16 GEMMs + output
transform) vs the direct implicit GEMM.  TF = executed MFMA TFLOP/s of the Winograd forms
(16/36 of the nominal FLOPs), nominal for the direct kernel."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
N = int(os.environ.get('MB_BATCH', '17'))
LAYERS = [('conv1b', 480, 640, 64, 64), ('conv2b', 240, 320, 256, 256), ('conv3b', 120, 160, 512, 512),
          ('conv4b', 60, 80, 1024, 1024), ('conv5', 60, 80, 1024, 512), ('conv6', 60, 80, 512, 256),
          ('feat3', 240, 320, 32, 32), ('feat5', 120, 160, 64, 64)]   # flow-feature tower (KFNet.py:70-76)
ONLY = os.environ.get('MB_LAYERS', '')
FUSED_ONLY = os.environ.get('MB_FUSED_ONLY', '') == '1'
for (name, H, W, ci, co) in LAYERS:
    if ONLY and name not in ONLY.split(','):
        continue
    x = torch.randn(N * H * W * ci + (1 << 22), device='cuda')      # (+16 MiB: the plane-stride experiments read past the tensor)
    u = torch.randn(16 * co * ci, device='cuda') * 0.02
    y = torch.empty(N * H * W * co + 4, device='cuda')
    YOFF = 4 * int(os.environ.get('MB_Y_MISALIGN', '0'))   # 1: y 4 bytes off a 16-byte boundary -> the kernels' dword-store epilogue
    Mt = N * (H // 2) * (W // 2)
    F16 = os.environ.get('MB_F16', '') == '1'       # fp16 operands (BASELINE config 5): four-wave form only
    if F16:
        if not (co >= 128 and ci % 64 == 0):
            continue
        u = u.half()
    d = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=3, kw=3, stride=1, relu=1, wino_order=ORDER, wino_form=int(os.environ.get('MB_WINO_FORM', '0')),
                      config=int(os.environ.get('KFN_WINO_CFG', '0')), operand_dtype=_lib.OPERAND_F16 if F16 else _lib.OPERAND_F32)
    t_fused = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd_fused(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr() + YOFF, st), 'wf'))
    fl = 2.0 * 16 * Mt * ci * co
    t_gemm = t_out = float('nan')
    if ci >= 128 and not FUSED_ONLY:
        ws = torch.empty(16 * Mt * co, device='cuda')
        t_gemm = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), ws.data_ptr(), 1, st), 'w'))
        t_out = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd(C.byref(d), x.data_ptr(), u.data_ptr(), None, y.data_ptr(), ws.data_ptr(), 2, st), 'w'))
        del ws
    dd = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=3, kw=3, stride=1, relu=1,
                       operand_dtype=_lib.OPERAND_F16 if F16 else _lib.OPERAND_F32, wino_order=int(os.environ.get('MB_F43_ORDER', ORDER)),
                       wino_form=int(os.environ.get('MB_F43_FORM', '0')))   # 2 four waves (wino4_kernel), 3 eight waves (wino4b_kernel)
    w9 = torch.randn(co * 9 * ci, device='cuda') * 0.02
    if F16:
        w9 = w9.half()
    t_dir = float('nan') if (FUSED_ONLY and not F16) else timeit(lambda: _lib.check(lib.kfn_conv2d_nhwc(C.byref(dd), x.data_ptr(), w9.data_ptr(), None, y.data_ptr(), st), 'c'))
    f43 = ''
    if os.environ.get('MB_F43', '1') == '1' and not F16 and lib.kfn_winograd_f43_supported(C.byref(dd)) == 1:
        u4 = torch.randn(36 * co * ci, device='cuda') * 0.02
        # MB_LAYOUT: 'xy' digits, 1 = KFN_LAYOUT_C16 (channel-blocked input / output; timing only -- the buffers hold noise either way)
        lay = os.environ.get('MB_LAYOUT', '00')
        dd.x_layout, dd.y_layout = int(lay[0]), int(lay[1])
        t4 = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd_f43(C.byref(dd), x.data_ptr(), u4.data_ptr(), None, y.data_ptr(), st), 'w4'))
        dd.x_layout, dd.y_layout = 0, 0
        M4 = N * (-(-H // 4)) * (-(-W // 4))
        f43 = ' | F(4x4,3x3) %.3f ms (%.1f TF exec, %.2fx the F(2x2,3x3) kernel)' % (t4, 2.0 * 36 * M4 * ci * co / t4 / 1e9, t_fused / t4)
        del u4
    print('%-7s %3dx%3d C%4d->%4d: FUSED %.3f ms (%.1f TF exec) | two-kernel %.3f ms = gemm %.3f (%.1f TF exec) + out %.3f | direct %.3f ms (%.1f TF)'
          % (name, H, W, ci, co, t_fused, fl / t_fused / 1e9, t_gemm + t_out, t_gemm, fl / t_gemm / 1e9, t_out, t_dir, fl * 2.25 / t_dir / 1e9) + f43, flush=True)
    del x, u, y, w9
