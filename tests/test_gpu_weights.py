"""-m gpu: weight import exercised ON the GPU and TF-default initialisation as a tested case.

The reference restores `ScoreNet/*` and `Temporal/*` from a checkpoint picked in --model_folder
(KFNet/train.py:317-321, KFNet/eval.py:66-68, tools/io.py:185-196) or fills a network from a numpy dict
{op_name: {param_name: array}} (cnn_wrapper/network.py:60-75); its untrained variables are glorot-uniform kernels with
zero biases (tf.layers defaults, cnn_wrapper/network.py:126-135).  Here: the .npz container keyed by TF variable names
through `kfnet_amd.KFNet.eval.main(['--model_folder', ...])`, the dict format through `Network.load`, and a sequence with
`synthetic_weights(init='glorot', bias_scale=0, flow_gain=1)` -- each against the CPU oracle on the same inputs."""
import os

import numpy as np
import pytest

from oracle import kfnet_oracle as O
from oracle import kfnet_oracle_torch as OT

pytestmark = pytest.mark.gpu

COORD_TOL = 1e-4     # BASELINE north_star: max-abs on the scene-coordinate channels
CONF_RTOL = 1e-4     # confidence 1/sigma: relative (SURVEY 7 'Tolerance on channel 3')


def _errs(rec, ref):
    dc = float(np.abs(rec[..., :3] - ref[..., :3]).max())
    dr = float((np.abs(rec[..., 3] - ref[..., 3]) / np.abs(ref[..., 3])).max())
    return dc, dr


def _write_input_folder(folder, imgs, transform_txt):
    from PIL import Image
    os.makedirs(folder)
    paths = []
    for i in range(imgs.shape[0]):
        p = os.path.join(folder, 'frame-%06d.color.png' % i)
        Image.fromarray(imgs[i]).save(p)
        paths.append(p)
    with open(os.path.join(folder, 'image_list.txt'), 'w') as f:
        f.write('\n'.join(paths) + '\n')
    np.savetxt(os.path.join(folder, 'transform.txt'), transform_txt)


def test_eval_main_restores_the_newest_snapshot_of_model_folder(tmp_path, capsys):
    """--model_folder -> get_snapshot (highest step wins, tools/io.py:185-196) -> load_npz -> engine: the records written
    for a PNG sequence equal the oracle's run with THAT snapshot's weights (and not the older snapshot's)."""
    from kfnet_amd.KFNet import eval as KE
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import save_npz, synthetic_weights
    H, W, T = 64, 96, 5
    model = tmp_path / 'model'
    model.mkdir()
    W_old, W_new = synthetic_weights(77), synthetic_weights(4321)
    save_npz(str(model / 'kfnet_weights-100.npz'), W_old)
    save_npz(str(model / 'kfnet_weights-2500.npz'), W_new)
    imgs = synthetic_sequence(T, H, W, seed=6)
    M = synthetic_transform()
    inp, out = tmp_path / 'in', tmp_path / 'out'
    _write_input_folder(str(inp), imgs, M)
    out.mkdir()
    rc = KE.main(['--input_folder', str(inp), '--output_folder', str(out), '--model_folder', str(model),
                  '--scene', 'heads', '--height', str(H), '--width', str(W), '--batch', '2'])
    assert rc == 0
    got = np.stack([np.load(out / ('coord_%d.npy' % i)) for i in range(T)])
    assert got.dtype == np.float32 and got.shape == (T, H // 8, W // 8, 4)
    T4 = KE.get_transform(str(inp / 'transform.txt'))          # what main() used (KFNet/train.py:49-58)
    ref = O.eval_sequence(imgs, W_new, T4, reset_period=500, dtype=np.float64)
    dc, dr = _errs(got, ref)
    print('model_folder run vs oracle: coord max-abs %.3g, conf max-rel %.3g' % (dc, dr))
    assert dc <= COORD_TOL and dr <= CONF_RTOL, (dc, dr)
    ref_old = O.eval_sequence(imgs[:2], W_old, T4, reset_period=500, dtype=np.float64)
    assert np.abs(got[:2, ..., :3] - ref_old[..., :3]).max() > 100 * COORD_TOL      # the step-100 snapshot was NOT used
    # an empty folder is reported, not papered over with random weights
    empty = tmp_path / 'empty'
    empty.mkdir()
    assert KE.main(['--input_folder', str(inp), '--output_folder', str(out), '--model_folder', str(empty),
                    '--scene', 'heads', '--height', str(H), '--width', str(W)]) == 1


def _load_dict(W, scope, names=None):
    """{op_name: {param_name: array}} of one variable scope (cnn_wrapper/network.py:60-75)."""
    table = {}
    for key, arr in W.items():
        sc, op, param = key.split('/')
        if sc == scope and (names is None or op in names):
            table.setdefault(op, {})[param] = arr
    return table


def test_network_load_fills_scoordnet_and_the_temporal_networks(tmp_path):
    """`Network.load` (numpy dict format) on networks built under variable_scope('ScoreNet') / ('Temporal'): an engine
    built with OTHER weights and then re-filled through SCoordNet.load / the feature tower's load / OFlowNet.load runs to
    the oracle's records for the loaded weights; a file entry without a variable raises unless ignore_missing."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.graph import variable_scope
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    H, W, T = 64, 96, 4
    W_a, W_b = synthetic_weights(11), synthetic_weights(2024)
    imgs = synthetic_sequence(T, H, W, seed=9)
    T4 = O.get_transform(synthetic_transform())
    eng = KFNetEngine(W_a, image_size=(H, W), batch=2, transform=T4, reset_period=500, max_chunk=8)
    rec_a = eng.process(eng.upload_frames(imgs)).cpu().numpy().copy()
    feat_names = set('feat%d' % i for i in range(1, 8))
    files = {}
    for tag, table in (('score', _load_dict(W_b, 'ScoreNet')),
                       ('feat', _load_dict(W_b, 'Temporal', feat_names)),
                       ('oflow', {k: v for k, v in _load_dict(W_b, 'Temporal').items() if k not in feat_names})):
        files[tag] = str(tmp_path / (tag + '.npy'))
        np.save(files[tag], table, allow_pickle=True)
    with variable_scope('ScoreNet'):
        eng.net.scoordnet.load(files['score'])
    with variable_scope('Temporal'):
        eng.net.feat_tower.load(files['feat'])
        eng.net.oflownet.load(files['oflow'])
    rec_b = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    ref_b = O.eval_sequence(imgs, W_b, T4, reset_period=500, dtype=np.float64)
    dc, dr = _errs(rec_b, ref_b)
    print('Network.load run vs oracle: coord max-abs %.3g, conf max-rel %.3g' % (dc, dr))
    assert dc <= COORD_TOL and dr <= CONF_RTOL, (dc, dr)
    assert np.abs(rec_b[..., :3] - rec_a[..., :3]).max() > 100 * COORD_TOL          # the weights really changed
    # network.py:69-75: an entry that names no variable raises ... unless ignore_missing
    bad = str(tmp_path / 'bad.npy')
    np.save(bad, {'conv99': {'kernel': np.zeros((3, 3, 4, 4), np.float32)}}, allow_pickle=True)
    with variable_scope('ScoreNet'):
        with pytest.raises(ValueError):
            eng.net.scoordnet.load(bad)
        eng.net.scoordnet.load(bad, ignore_missing=True)
    # ... and a wrong shape is never accepted
    wrong = str(tmp_path / 'wrong.npy')
    np.save(wrong, {'conv7': {'kernel': np.zeros((1, 1, 256, 64), np.float32)}}, allow_pickle=True)
    with variable_scope('ScoreNet'):
        with pytest.raises(ValueError):
            eng.net.scoordnet.load(wrong)


@pytest.mark.parametrize('size,frames,batch', [((64, 96), 7, 2), ((480, 640), 3, 3)])
def test_tf_default_initialisation_glorot_zero_bias(size, frames, batch):
    """What an untrained TF graph holds: glorot-uniform kernels, zero biases, no flow gain.
    Activations shrink through the 12 ReLU layers, so the outputs are small and the bar is RELATIVE: max-abs error <=
    1e-4 of the reference's range per channel group (coordinates; confidence relative per pixel as everywhere)."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    Wg = synthetic_weights(1234, init='glorot', bias_scale=0.0, flow_gain=1.0)
    assert all(not v.any() for k, v in Wg.items() if k.endswith('/bias'))
    H, W = size
    imgs = synthetic_sequence(frames, H, W, seed=1)
    T4 = O.get_transform(synthetic_transform())
    full = H * W > 64 * 96      # full size: the torch fp32 restatement (the fp64 numpy one takes minutes per frame)
    # (raw network outputs first -- identity transform: the rigid transform's translation would hide how small they are)
    I4 = np.eye(4, dtype=np.float32)
    eng = KFNetEngine(Wg, image_size=size, batch=batch, transform=I4, reset_period=500, max_chunk=8, emit_debug=True)
    rec = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    if full:
        ref = OT.eval_sequence(imgs, Wg, I4, reset_period=500)
    else:
        ref, dbg = O.eval_sequence(imgs, Wg, I4, reset_period=500, dtype=np.float64, return_debug=True)
    scale = float(np.abs(ref[..., :3]).max())
    dc, dr = _errs(rec, ref)
    print('glorot %dx%d: coord range %.3g, max-abs %.3g (%.3g of range), conf max-rel %.3g' % (H, W, scale, dc, dc / scale, dr))
    assert scale > 0 and dc <= COORD_TOL * max(scale, 1e-3), (dc, scale)
    assert dc <= COORD_TOL and dr <= CONF_RTOL, (dc, dr)
    if not full:
        flow_ref = np.stack([dbg[t]['flow'][0] for t in range(1, frames)])
        flow = eng.debug(frames)['flow'][1:]
        df = float(np.abs(flow - flow_ref).max())
        print('glorot %dx%d: flow max-abs %.3g px, flow range [%.3f, %.3f]' % (H, W, df, flow_ref.min(), flow_ref.max()))
        assert df <= 1e-4
        # with a (near-)uniform softmax the flow is the soft-argmax bias of SURVEY App. E7: about (-0.5, -0.5)
        assert np.abs(flow_ref + 0.5).max() < 0.5
    # and through the transform as eval.py applies it
    eng2 = KFNetEngine(Wg, image_size=size, batch=batch, transform=T4, reset_period=500, max_chunk=8)
    rec2 = eng2.process(eng2.upload_frames(imgs)).cpu().numpy()
    ref2 = (OT.eval_sequence(imgs, Wg, T4, reset_period=500) if full
            else O.eval_sequence(imgs, Wg, T4, reset_period=500, dtype=np.float64))
    dc2, dr2 = _errs(rec2, ref2)
    assert dc2 <= COORD_TOL and dr2 <= CONF_RTOL, (dc2, dr2)
