"""Microbenchmark of SCoordNet's wide layers at BASELINE config 5's geometry (540x960, fp16 MFMA operands):
fp32 activations in memory (round 2's form: direct kernel / four-wave Winograd) against fp16 activations
(kfn_conv_desc.x_dtype = y_dtype = KFN_ACT_F16) over the tile / k-step variants.  TF = nominal direct-conv TFLOP/s.

    MB_BATCH=16 python tools/mb_f16.py [layer ...]
"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kfnet_amd import _lib
if os.environ.get('MB_LIB'):          # a timing-experiment build (tools/mb/build_hot.sh)
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
N = int(os.environ.get('MB_BATCH', '16'))
H0, W0 = int(os.environ.get('MB_H', '540')), int(os.environ.get('MB_W', '960'))
def lvl(v, k):
    for _ in range(k): v = -(-v // 2)
    return v
LAYERS = [('conv1b', 0, 64, 64, 3, 1), ('conv2a', 0, 64, 256, 3, 2), ('conv2b', 1, 256, 256, 3, 1),
          ('conv3a', 1, 256, 512, 3, 2), ('conv3b', 2, 512, 512, 3, 1), ('conv4a', 2, 512, 1024, 3, 2),
          ('conv4b', 3, 1024, 1024, 3, 1), ('conv5', 3, 1024, 512, 3, 1), ('conv6', 3, 512, 256, 3, 1),
          ('conv7', 3, 256, 128, 1, 1)]
only = sys.argv[1:]
for (name, lv, ci, co, k, s) in LAYERS:
    if only and name not in only:
        continue
    H, W = lvl(H0, lv), lvl(W0, lv)
    Ho, Wo = -(-H // s), -(-W // s)
    x32 = torch.randn(N * H * W * ci, device='cuda')
    x16 = x32.half()
    w = (torch.randn(co * k * k * ci, device='cuda') * 0.02).half()
    y32 = torch.empty(N * Ho * Wo * co, device='cuda')
    y16 = torch.empty(N * Ho * Wo * co, device='cuda', dtype=torch.float16)
    fl = 2.0 * N * Ho * Wo * k * k * ci * co
    res = []
    def conv(xd, yd, cfg, ks, xt, yt, wpath=0):
        d = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=k, kw=k, stride=s, relu=1,
                          config=cfg, operand_dtype=_lib.OPERAND_F16, x_dtype=xt, y_dtype=yt, k_step=ks, weights_path=wpath)
        return timeit(lambda: _lib.check(lib.kfn_conv2d_nhwc(C.byref(d), xd.data_ptr(), w.data_ptr(), None, yd.data_ptr(), st), 'c'))
    t = conv(x32, y32, 0, 0, 0, 0)
    res.append('f32act direct %.3f ms %4.0f TF' % (t, fl / t / 1e9))
    if k == 3 and s == 1 and co >= 128 and ci % 64 == 0:
        u = (torch.randn(16 * co * ci, device='cuda') * 0.02).half()
        d = _lib.ConvDesc(N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, cout_pad=co, ldy=co, kh=3, kw=3, stride=1, relu=1,
                          operand_dtype=_lib.OPERAND_F16)
        t = timeit(lambda: _lib.check(lib.kfn_conv2d_winograd_fused(C.byref(d), x32.data_ptr(), u.data_ptr(), None, y32.data_ptr(), st), 'w'))
        res.append('f32act wino3 %.3f ms %4.0f TF' % (t, fl / t / 1e9))
        del u
    cfgs = [(2, '128x128'), (9, '128x256'), (13, '256x256'), (14, '256x256w8')] if co >= 128 else [(7, '192x64'), (3, '128x64'), (12, '256x64'), (15, '512x64')]
    if os.environ.get('MB_CFGS'):
        cfgs = [c for c in cfgs if str(c[0]) in os.environ['MB_CFGS'].split(',')]
    for cfg, cn in cfgs:
        for ks in ((16, 32) if (ci % 64 == 0 and not os.environ.get('MB_K16_ONLY')) else (16,)):
            t = conv(x16, y16, cfg, ks, 1, 1, 1)
            res.append('f16act %s k%d %.3f ms %4.0f TF' % (cn, ks, t, fl / t / 1e9))
            if ks == 16 and cfg in (2, 9, 12, 13, 14, 15):
                if cfg != 12:
                    t = conv(x16, y16, cfg, ks, 1, 1, 2)
                    res.append('  +B via LDS-DMA %.3f ms %4.0f TF' % (t, fl / t / 1e9))
                t = conv(x16, y16, cfg, ks, 1, 1, 3)
                res.append('  +A,B via LDS-DMA %.3f ms %4.0f TF' % (t, fl / t / 1e9))
    print('%-6s %3dx%3d C%4d->%4d s%d: ' % (name, H, W, ci, co, s) + ' | '.join(res), flush=True)
    del x32, x16, w, y32, y16
