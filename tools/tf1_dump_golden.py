#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Reference-side golden-vector hook: dump what the ORIGINAL zlthinker/KFNet (Python 2.7 +
TensorFlow 1.10-1.13) computes for the seeded inputs of this repository's parity fixtures.

THIS FILE IS NEVER IMPORTED OR RUN BY kfnet_amd, ITS TESTS OR bench.py.  TensorFlow 1.x cannot be
installed where this repository is built, so the repository's oracle is
"parity unpinned".  Whoever holds a TF1 environment can pin it:

  1. (Python 3, this repository)   python tests/golden/make_tf1_inputs.py
         -> tests/golden/tf1_inputs_small.npz (gitignored: ~100 MB of seeded weights, the
            uint8 frames, the 4x4 transform, reset period)
  2. (Python 2.7 + TF1, a checkout of zlthinker/KFNet)
         python tools/tf1_dump_golden.py --reference /path/to/KFNet \
                --inputs tests/golden/tf1_inputs_small.npz --out tests/golden/tf1_kfnet_small.npz
  3. commit tests/golden/tf1_kfnet_small.npz (a few hundred KB).  tests/test_golden.py picks every
     tests/golden/tf1_*.npz up automatically and holds the oracle AND the HIP path to it.

What it does: builds the reference's own graph -- KFNet(images[2,H,W,3], spec, False, False),
GetMeasureCoord2, GetKFCoordRecursive(last_coord, last_uncertainty), GetNIS, ApplyTransform, the
exact calls of KF_fusion (KFNet/train.py:241-266) minus the queue-runner input pipeline, which
is replaced by placeholders -- assigns the seeded weights to the variables by name
(ScoreNet/*, Temporal/*), then walks the frames with eval.py's host loop (KFNet/eval.py:77-126:
pair schedule [[1,0],[0,1],[1,2],...], reset at i % reset_period == 0, raw KF state fed back,
optional NIS gate on the output, record = concat(T.x, 1/sigma)).

Written for Python 2.7 / TF 1.x (no f-strings, no keyword-only arguments); it also runs on
Python 3 with tensorflow.compat.v1 if tf.contrib is available.
"""
from __future__ import print_function

import argparse
import os
import sys

import numpy as np


def build(reference, H, W):
    sys.path.insert(0, reference)
    sys.path.insert(0, os.path.join(reference, 'KFNet'))
    import tensorflow as tf
    import KFNet as ref_kfnet            # KFNet/KFNet.py of the reference (implicit relative imports)
    from util import ApplyTransform      # KFNet/util.py

    spec = ref_kfnet.KFNetDataSpec()
    spec.batch_size = 2                  # KFNet/eval.py:41
    spec.image_size = [H, W]
    spec.crop_size = [H, W]
    h, w = H // 8, W // 8                # KFNet/eval.py:49-50 (the fixtures use sizes divisible by 8)
    images = tf.placeholder(tf.float32, [2, H, W, 3], name='images')
    last_coord = tf.placeholder(tf.float32, [1, h, w, 3], name='last_coord_in')
    last_unc = tf.placeholder(tf.float32, [1, h, w, 1], name='last_uncertainty_in')
    transform = tf.placeholder(tf.float32, [4, 4], name='transform_in')

    net = ref_kfnet.KFNet(images, spec, False, False)                  # train.py:245
    m_coord, m_unc = net.GetMeasureCoord2()                            # train.py:247
    t_coord, t_unc, kf_coord, kf_unc = net.GetKFCoordRecursive(last_coord, last_unc)   # train.py:248-249
    nis = net.GetNIS(m_coord, m_unc, t_coord, t_unc)                   # train.py:250
    fetch = dict(z=m_coord, sz=m_unc, temp_x=t_coord, temp_s=t_unc, kf_x=kf_coord, kf_s=kf_unc, nis=nis,
                 t_z=ApplyTransform(m_coord, transform), t_temp=ApplyTransform(t_coord, transform),
                 t_kf=ApplyTransform(kf_coord, transform), feat=net.temp_feat_maps)
    g = tf.get_default_graph()
    for key, name in (('flow', 'flow:0'), ('prob', 'prob_reshape:0')):  # KFNet/KFNet.py:381-385
        try:
            fetch[key] = g.get_tensor_by_name(name)
        except (KeyError, ValueError):
            print('warning: tensor %s not found; %s will be missing from the dump' % (name, key))
    return tf, dict(images=images, last_coord=last_coord, last_unc=last_unc, transform=transform), fetch


def assign_weights(tf, sess, weights):
    """weights: {tf variable name without ':0': ndarray} (kfnet_amd/weights.py container)."""
    todo = dict(weights)
    ops = []
    for v in tf.global_variables():
        name = v.name[:-2]
        if name in todo:
            arr = todo.pop(name)
            assert tuple(v.shape.as_list()) == tuple(arr.shape), (name, v.shape, arr.shape)
            ops.append(v.assign(arr))
        else:
            print('warning: variable %s has no seeded value' % name)
    if todo:
        raise KeyError('weights without a TF variable: %s' % sorted(todo)[:5])
    sess.run(ops)


def run_sequence(tf, sess, ph, fetch, frames, T4, reset_period, nis_gate):
    """KFNet/eval.py:77-126 on in-memory frames."""
    n = frames.shape[0]
    h, w = ph['last_coord'].shape.as_list()[1:3]
    state_x = np.zeros((1, h, w, 3), np.float32)     # eval.py: glorot garbage, overwritten at step 0
    state_s = np.ones((1, h, w, 1), np.float32)
    records, stages = [], []
    for i in range(n):
        pair = (1, 0) if i == 0 else (i - 1, i)       # KFNet/train.py:67-71
        batch = np.stack([frames[pair[0]], frames[pair[1]]]).astype(np.float32)
        o = sess.run(fetch, {ph['images']: batch, ph['last_coord']: state_x, ph['last_unc']: state_s,
                             ph['transform']: T4})
        t_kf = o['t_kf']
        if nis_gate:                                   # eval.py:87-92
            mask = (np.sum(o['nis'], axis=-1) > 7.815).astype(np.float32)[..., None]
            t_kf = mask * o['t_z'] + (1.0 - mask) * t_kf
        kf_s = o['kf_s']
        if i % reset_period == 0:                      # eval.py:94-101
            state_x, state_s = o['z'], o['sz']
            t_kf, kf_s = o['t_z'], o['sz']
        else:                                          # eval.py:103-104
            state_x, state_s = o['kf_x'], o['kf_s']
        records.append(np.concatenate([t_kf[-1], 1.0 / kf_s[-1]], axis=-1).astype(np.float32))   # eval.py:123
        stages.append(o)
    return np.stack(records), stages


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--reference', required=True, help='root of a zlthinker/KFNet checkout')
    ap.add_argument('--inputs', required=True, help='npz written by tests/golden/make_tf1_inputs.py')
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    z = np.load(a.inputs)
    frames = z['images']
    T4 = z['transform'].astype(np.float32)
    reset_period = int(z['reset_period'])
    weights = dict((k[len('w:'):], z[k]) for k in z.files if k.startswith('w:'))
    H, W = frames.shape[1:3]
    tf, ph, fetch = build(a.reference, H, W)
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        assign_weights(tf, sess, weights)
        rec, st = run_sequence(tf, sess, ph, fetch, frames, T4, reset_period, False)
        rec_nis, _ = run_sequence(tf, sess, ph, fetch, frames, T4, reset_period, True)
    d = st[1]                                           # stage outputs of step 1 = pair (0, 1)
    hw = d['z'].shape[1:3]
    out = dict(images=frames, transform=T4, records=rec, records_nis=rec_nis, reset_period=reset_period,
               seed_w=int(z['seed_w']), seed_img=int(z['seed_img']), weights_sha256=z['weights_sha256'],
               z1=d['z'], sz1=d['sz'], feat1=d['feat'], temp_x1=d['temp_x'], temp_s1=d['temp_s'], nis1=d['nis'],
               source='tensorflow %s, reference checkout %s' % (tf.__version__, a.reference))
    if 'flow' in d:
        out['flow1'] = d['flow']
    if 'prob' in d:
        out['prob1'] = d['prob'].reshape(hw[0] * hw[1], -1)
    np.savez_compressed(a.out, **out)
    print('wrote', a.out)


if __name__ == '__main__':
    main()
