#!/usr/bin/env python
"""Calibration of the error model tests/test_gpu_ops.py holds the convolution kernels to (Next #3) -- CPU only.

Model: a convolution output y = sum_k a_k b_k over K = kh*kw*Cin products has RMS magnitude S = sqrt(K) * rms(x) * rms(w)
(zero-mean weights); an fp32 evaluation in any order is off by a small multiple of eps32 * S, times the gain of the
minimal-filtering transforms where they are used (F(2x2,3x3): B^T / A^T rows sum |.| <= 2 / 3; F(4x4,3x3): up to 10 / 8).
This script measures err_max / (eps32 * S) for a direct fp32 convolution, an fp32 F(2x2,3x3) and an fp32 F(4x4,3x3) evaluation
(everything the GPU does in fp32 done in fp32 here, U prepared in fp64 -- tools/experiments/f43_error_budget.py's emulation)
on the op tests' own shapes and data distribution, so the constants in tests/test_gpu_ops.py::conv_err_bound can be read off.

    python tools/experiments/conv_error_model.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from f43_error_budget import tf_same_conv, toom_cook, wino_conv  # noqa: E402

EPS = 2.0 ** -24
CASES = [(2, 32, 16, 16, 64), (3, 60, 80, 32, 64), (2, 30, 40, 64, 128), (1, 68, 120, 48, 72), (3, 29, 35, 32, 100),
         (2, 32, 16, 512, 64), (1, 60, 80, 1024, 128), (1, 60, 80, 256, 128), (2, 12, 16, 512, 128)]


def main():
    torch.set_num_threads(8)
    m23 = toom_cook(2, 3, [0, 1, -1])
    m43 = toom_cook(4, 3, [0, 1, -1, 2, -2])
    print('%-26s %9s | %21s | %21s | %21s' % ('case', 'S', 'direct err  /epsS', 'F(2x2) err  /epsS', 'F(4x4) err  /epsS'))
    for case in CASES:
        n, h, w, ci, co = case
        rng = np.random.default_rng(n * 1000 + h * 10 + ci + 43)
        x = np.maximum(rng.normal(size=(n, h, w, ci)), 0).astype(np.float32)
        wt = (rng.normal(size=(3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
        b = rng.normal(size=co).astype(np.float32)
        S = np.sqrt(9 * ci) * np.sqrt(np.mean(x.astype(np.float64) ** 2)) * np.sqrt(np.mean(wt.astype(np.float64) ** 2))
        xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
        wtt = torch.from_numpy(wt).permute(3, 2, 0, 1).contiguous()
        bt = torch.from_numpy(b)
        ref = tf_same_conv(xt.double(), wtt.double(), bt.double(), 1, False)
        out = []
        y = tf_same_conv(xt, wtt, bt, 1, False)
        out.append(float((y.double() - ref).abs().max()))
        for m, mats in ((2, m23), (4, m43)):
            errs = []
            for i in range(n):
                y = wino_conv(xt[i:i + 1], wtt, bt, False, m, mats)
                errs.append(float((y.double() - ref[i:i + 1]).abs().max()))
            out.append(max(errs))
        print('%-26s %9.3f | %10.3g %10.1f | %10.3g %10.1f | %10.3g %10.1f'
              % (case, S, out[0], out[0] / (EPS * S), out[1], out[1] / (EPS * S), out[2], out[2] / (EPS * S)))


if __name__ == '__main__':
    main()
