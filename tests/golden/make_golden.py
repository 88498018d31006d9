#!/usr/bin/env python
"""Generates tests/golden/kfnet_small.npz from the numpy fp64 oracle.

The reference (TF-1.x, Python 2) cannot be imported or run here and ships
no golden vectors (F4), so these fixtures pin the build's OWN oracle -- parity is
"unpinned" with respect to TensorFlow.  Inputs: 5 seeded uint8 frames 64x96, the seeded
synthetic weights (regenerated from the seed; only a checksum is stored), a seeded rigid
transform, reset_period 4.  Outputs: per-frame records and the stage outputs of frame 1.

    python tests/golden/make_golden.py           (kfnet_small.npz, 64x96)
    python tests/golden/make_golden.py --full    (kfnet_full.npz: a 4-frame 480x640 sequence, records only;
                                                  frame 0 is the single-frame case = pure measurement)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from kfnet_amd.synth import synthetic_sequence, synthetic_transform  # noqa: E402
from kfnet_amd.weights import synthetic_weights, variable_specs  # noqa: E402
from oracle import kfnet_oracle as O  # noqa: E402

SEED_W, SEED_IMG, RESET = 1234, 1, 4


def weights_digest(W):
    h = hashlib.sha256()
    for n, _, _ in variable_specs():
        h.update(W[n + '/kernel'].tobytes())
        h.update(W[n + '/bias'].tobytes())
    return h.hexdigest()


def main_full():
    """Full-size fixture: 4 frames 480x640 of the seed-1 stream through the numpy
    fp64 gold (about 40 s of CPU); only the fp32 [4,60,80,4] records and the seeds are stored."""
    W = synthetic_weights(SEED_W)
    imgs = synthetic_sequence(4, 480, 640, seed=SEED_IMG)
    T4 = O.get_transform(synthetic_transform())
    rec = O.eval_sequence(imgs, W, T4, reset_period=500, dtype=np.float64)
    out = dict(records=rec.astype(np.float32), transform=T4, weights_sha256=np.array(weights_digest(W)),
               images_sha256=np.array(hashlib.sha256(imgs.tobytes()).hexdigest()),
               seed_w=SEED_W, seed_img=SEED_IMG, reset_period=500, frames=4, height=480, width=640)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'kfnet_full.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), 'bytes')


def main():
    if '--full' in sys.argv:
        return main_full()
    W = synthetic_weights(SEED_W)
    imgs = synthetic_sequence(5, 64, 96, seed=SEED_IMG)
    T4 = O.get_transform(synthetic_transform())
    rec, dbg = O.eval_sequence(imgs, W, T4, reset_period=RESET, dtype=np.float64, return_debug=True)
    rec_nis = O.eval_sequence(imgs, W, T4, reset_period=RESET, nis_gate=True, dtype=np.float64)
    d = dbg[1]
    out = dict(images=imgs, transform=T4, records=rec, records_nis=rec_nis,
               z1=d['z'].astype(np.float32), sz1=d['sz'].astype(np.float32), feat1=d['feat'].astype(np.float32),
               prob1=d['prob'].astype(np.float32), sigma_trans1=d['sigma_trans'].astype(np.float32),
               flow1=d['flow'].astype(np.float32), temp_x1=d['temp_x'].astype(np.float32),
               temp_s1=d['temp_s'].astype(np.float32), nis1=d['nis'].astype(np.float32),
               weights_sha256=np.array(weights_digest(W)), seed_w=SEED_W, seed_img=SEED_IMG, reset_period=RESET)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'kfnet_small.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
