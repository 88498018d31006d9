#!/usr/bin/env python
"""Step 1 of pinning the oracle to TensorFlow (see tools/tf1_dump_golden.py): write the seeded
inputs of the small fixture -- frames, transform, reset period and ALL seeded weights under
their TF variable names ('w:<name>') -- to tests/golden/tf1_inputs_small.npz (about 100 MB, not
committed: *.npz inputs are regenerated from the seeds).

    python tests/golden/make_tf1_inputs.py [--full]      (--full: 480x640 frames instead of 64x96)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from kfnet_amd.synth import synthetic_sequence, synthetic_transform  # noqa: E402
from kfnet_amd.weights import synthetic_weights  # noqa: E402
from make_golden import RESET, SEED_IMG, SEED_W, weights_digest  # noqa: E402


def main():
    full = '--full' in sys.argv
    W = synthetic_weights(SEED_W)
    imgs = synthetic_sequence(5, 480 if full else 64, 640 if full else 96, seed=SEED_IMG)
    T4 = np.linalg.inv(synthetic_transform())
    out = dict(images=imgs, transform=T4, reset_period=RESET, seed_w=SEED_W, seed_img=SEED_IMG,
               weights_sha256=np.array(weights_digest(W)))
    for k, v in W.items():
        out['w:' + k] = v
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        'tf1_inputs_%s.npz' % ('full' if full else 'small'))
    np.savez(path, **out)
    print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
