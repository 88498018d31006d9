"""Microbenchmark: MFMA conv kernel efficiency vs K (1x1 conv, fixed M and N): isolates the
per-tile fixed cost (pipeline fill/drain, setup, epilogue) from everything else."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
M = 326400
for N in (256, 1024):
    for K in (128, 256, 512, 1024, 2048, 4096):
        x = torch.randn(M * K, device='cuda')
        w = torch.randn(N * K, device='cuda') * 0.02
        y = torch.empty(M * N, device='cuda')
        for cfg in (1, 2):
            d = _lib.ConvDesc(N=M // 256, H=16, W=16, Cin=K, ldx=K, Cout=N, cout_pad=N, ldy=N, kh=1, kw=1, stride=1, config=cfg)
            t = timeit(lambda: _lib.check(lib.kfn_conv2d_nhwc(C.byref(d), x.data_ptr(), w.data_ptr(), None, y.data_ptr(), st), 'c'))
            print('N=%4d K=%4d cfg%d: %.3f ms  %.1f TF   (in %.2f GB, out %.2f GB -> %.2f TB/s)' % (
                N, K, cfg, t, 2.0 * M * N * K / t / 1e9, M * K * 4 / 1e9, M * N * 4 / 1e9, (M * K + M * N) * 4 / t / 1e9))
        del x, w, y
