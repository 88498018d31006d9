#!/usr/bin/env python
"""Runs ONLY the roofline shape of the Kalman scan (S=256 sequences x T=64 frames x 60x80 px per
launch, bench.py's `roofline_kalman`) so that a rocprofv3 PMC pass samples exactly that launch:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- python tools/kalman_roofline.py

Expected HBM traffic per launch: 28 B/px in + 16 B/px out = 44 B x 256 x 64 x 4800 px = 3.46 GB
(+ 39 MB of state load/store), against 5.98 GB of algorithmic traffic (76 B/px)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extra as bench  # noqa: E402

if __name__ == '__main__':
    # `--T 256`: SURVEY 8(d)'s default shape (S = 256 x T = 256) alone, for its own PMC passes (the kernel name is the same, so
    # the two shapes cannot share a pass: tools/pmc_traffic.py averages per kernel name)
    T = int(sys.argv[sys.argv.index('--T') + 1]) if '--T' in sys.argv else 64
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    # `--smooth-flow`: a per-frame displacement + 0.05 px of noise instead of an independent random neighbour per pixel (a side
    # measurement: bench.py's `roofline_kalman` keeps the random field, the worst case for the LDS gather)
    out = {'scan': bench.kalman_roofline(dev, T=T, flow='smooth' if '--smooth-flow' in sys.argv else 'random')}
    if T == 64:
        out['fuse'] = bench.kalman_fuse_roofline(dev)
    print(json.dumps(out))
