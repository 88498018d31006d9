#!/usr/bin/env python
"""Dump what is needed to analyse config 5's parity offline: for S sequences x T frames of 540x960,
the records and flows of the fp32 and of the fp16-operand HIP paths (+ the fp32 path's predicted sigma and
measurement sigma, from which the Kalman gain follows).

    python tools/c5_parity_dump.py gpurun_out/c5_parity.npz [S] [T]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kfnet_amd.engine import KFNetEngine  # noqa: E402
from kfnet_amd.synth import synthetic_sequence  # noqa: E402
from kfnet_amd.weights import synthetic_weights  # noqa: E402

out = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
W = synthetic_weights(1234)
seqs = np.stack([synthetic_sequence(T, 540, 960, seed=3 + s) for s in range(S)])
dev = torch.from_numpy(seqs).cuda()
T4 = np.eye(4, dtype=np.float32)
res = {}
for mode in ('f32', 'f16'):
    eng = KFNetEngine(W, image_size=(540, 960), batch=8, transform=T4, reset_period=500, max_chunk=S * T,
                      conv_operands=mode)
    res['rec_' + mode] = eng.process_sequences(dev).cpu().numpy().copy()
    d = eng.debug(S * T)
    res['flow_' + mode] = d['flow'].reshape(S, T, eng.h, eng.w, 2).copy()
    res['sigt_' + mode] = d['sigma_trans'].reshape(S, T, eng.h, eng.w).copy()
    res['meas_' + mode] = d['meas'].reshape(S, T, eng.h, eng.w, 4).copy()
    del eng
    torch.cuda.empty_cache()
np.savez_compressed(out, **res)
print('wrote', out, {k: v.shape for k, v in res.items()})
