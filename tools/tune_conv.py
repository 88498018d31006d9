"""Time every tile configuration for every MFMA conv launch of the B-frame KFNet graph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kfnet_amd import _lib
from kfnet_amd.engine import KFNetEngine
from kfnet_amd.graph import ConvOp, WinogradConvOp
from kfnet_amd.weights import synthetic_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 17
eng = KFNetEngine(synthetic_weights(1234), batch=B, max_chunk=B)
lib = eng.lib
g = eng.graph
g.active = (B, B)
st = eng._stream()
frames = torch.randint(0, 255, (B, 480, 640, 3), dtype=torch.uint8, device='cuda')
eng._set_batch_images(frames, 0, B, st)
g.run(st, eng.heavy_ops, active=(B, B)); g.active = (B, B)
torch.cuda.synchronize()
def t(op, reps=6):
    for _ in range(2): op.launch(lib, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): op.launch(lib, st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for op in eng.heavy_ops:
    if not isinstance(op, ConvOp) or op.epilogue == _lib.EPI_L2NORM: continue
    if len(sys.argv) > 2 and op.name not in sys.argv[2].split(','): continue
    res = {}
    for cfg in (0, 1, 2, 9, 8, 7, 3, 6, 4, 5):
        op.config = cfg
        try: res[cfg] = t(op)
        except _lib.KfnError: pass
    op.config = 0
    best = min((v, k) for k, v in res.items() if k)
    print('%-11s %-5s M=%8d Cin=%4d Cout=%4d  auto %.3f | best cfg %d %.3f | %s' % (
        op.name, 'wino' if isinstance(op, WinogradConvOp) else ('dec' if op.transposed else 'conv'),
        op.y.pixels, op.x.shape[3], op.y.shape[3], res[0], best[1], best[0],
        ' '.join('%d:%.3f' % (k, v) for k, v in res.items() if k)))
