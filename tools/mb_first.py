"""Microbenchmark of the first layer (kfn_first_conv_u8: conv1a 64 ch + feat1 16 ch from the uint8 image).
MB_LIB=<path> times another build of the library; prints ms per launch (back-to-back launches) and a checksum."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from kfnet_amd import _lib
if os.environ.get('MB_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MB_LIB'])
lib = _lib.load()
N, H, W = int(os.environ.get('MB_N', 32)), 480, 640
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.integers(0, 256, size=(N, H, W, 3), dtype=np.uint8)).cuda()
w1 = torch.from_numpy((rng.normal(size=(27, 64)) / 5).astype(np.float32)).cuda()
b1 = torch.from_numpy(rng.normal(size=64).astype(np.float32)).cuda()
w2 = torch.from_numpy((rng.normal(size=(27, 16)) / 5).astype(np.float32)).cuda()
b2 = torch.from_numpy(rng.normal(size=16).astype(np.float32)).cuda()
y1 = torch.empty(N * H * W * 64, device='cuda')
y2 = torch.empty(N * H * W * 16, device='cuda')
s = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(lib.kfn_first_conv_u8(img.data_ptr(), N, H, W, w1.data_ptr(), b1.data_ptr(), y1.data_ptr(), 64,
                                     w2.data_ptr(), b2.data_ptr(), y2.data_ptr(), 16, s), 'first')
for _ in range(5): run()
torch.cuda.synchronize()
best = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 20)
gb = (N * H * W * (3 + 80 * 4)) / 1e9
ms = float(np.median(best))
print('%s: %.4f ms (min %.4f)  %.2f TB/s  checksum %.6e %.6e' % (os.environ.get('MB_LIB', 'product'), ms, min(best), gb / ms,
      float(y1.double().sum()), float(y2.double().sum())))
