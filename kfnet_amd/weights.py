"""Weight container for the KFNet prediction path.

The reference restores two TF variable scopes from a tf.train.Saver checkpoint
(`RestoreFromScope(sess, snapshot, 'ScoreNet' | 'Temporal')`, KFNet/train.py:317-321,
KFNet/eval.py:66-68).  TF checkpoints cannot be read without TensorFlow, so this
framework's container is a flat ``{tf_variable_name: float32 ndarray}`` dict stored as
``.npz`` -- the keys and layouts are exactly the TF ones:

    ScoreNet/<layer>/kernel [kh,kw,Cin,Cout]   ScoreNet/<layer>/bias [Cout]
    Temporal/feat{1..7}/kernel|bias            Temporal/conv*/kernel|bias
    Temporal/upconv{2,1,0}/kernel [kh,kw,Cout,Cin] (conv2d_transpose layout)
    Temporal/{fc1,fc2,uncertainty}/kernel [in,out]

An offline converter (run where TensorFlow exists) only has to dump
``{v.name[:-2]: sess.run(v)}`` for these scopes with ``np.savez``.
`Network.load`'s ``{op_name: {param_name: array}}`` dict (cnn_wrapper/network.py:60-75)
maps 1:1: ``flat[scope + '/' + op_name + '/' + param_name]``.

`synthetic_weights` produces seeded random weights of the right architecture for
benchmarks and parity tests (no trained weights are reachable from this environment).
"""
import numpy as np

# (name, kind, shape)  -- generation order == SURVEY.md App. B order
def variable_specs():
    specs = []
    sc = [('conv1a', 3, 3, 64), ('conv1b', 3, 64, 64), ('conv2a', 3, 64, 256),
          ('conv2b', 3, 256, 256), ('conv3a', 3, 256, 512), ('conv3b', 3, 512, 512),
          ('conv4a', 3, 512, 1024), ('conv4b', 3, 1024, 1024), ('conv5', 3, 1024, 512),
          ('conv6', 3, 512, 256), ('conv7', 1, 256, 128), ('prediction', 1, 128, 4)]
    for n, k, ci, co in sc:  # cnn_wrapper/SCoordNet.py:21-32
        specs.append(('ScoreNet/' + n, 'conv', (k, k, ci, co)))
    ft = [('feat1', 3, 16), ('feat2', 16, 32), ('feat3', 32, 32), ('feat4', 32, 64),
          ('feat5', 64, 64), ('feat6', 64, 128), ('feat7', 128, 32)]
    for n, ci, co in ft:  # KFNet/KFNet.py:318-338
        specs.append(('Temporal/' + n, 'conv', (3, 3, ci, co)))
    of = [('conv0', 32, 32), ('conv1a', 32, 32), ('conv1b', 32, 32), ('conv2a', 32, 64),
          ('conv2b', 64, 64), ('conv3a', 64, 128), ('conv3b', 128, 128)]
    for n, ci, co in of:  # cnn_wrapper/OFlowNet.py:19-25
        specs.append(('Temporal/' + n, 'conv', (3, 3, ci, co)))
    specs.append(('Temporal/upconv2', 'deconv', (3, 3, 64, 128)))   # OFlowNet.py:26
    specs.append(('Temporal/conv4', 'conv', (3, 3, 128, 64)))       # :28-30
    specs.append(('Temporal/upconv1', 'deconv', (3, 3, 32, 64)))    # :31
    specs.append(('Temporal/conv5', 'conv', (3, 3, 64, 32)))        # :33-35
    specs.append(('Temporal/upconv0', 'deconv', (3, 3, 16, 32)))    # :36
    specs.append(('Temporal/conv6', 'conv', (3, 3, 48, 16)))        # :38-40
    specs.append(('Temporal/prediction', 'conv', (3, 3, 16, 1)))    # :41
    specs.append(('Temporal/fc1', 'dense', (128, 64)))              # :50-55
    specs.append(('Temporal/fc2', 'dense', (64, 32)))
    specs.append(('Temporal/uncertainty', 'dense', (32, 1)))
    return specs


def _fans(kind, shape):
    if kind == 'dense':
        return shape[0], shape[1]
    rf = shape[0] * shape[1]
    # TF computes fans from the variable shape: fan_in = shape[-2]*rf, fan_out = shape[-1]*rf
    return shape[2] * rf, shape[3] * rf


def synthetic_weights(seed=1234, init='he', bias_scale=0.05, flow_gain=8.0):
    """Seeded random weights keyed by TF variable names.

    init='glorot': tf.layers default (glorot-uniform kernels); with bias_scale=0 this is
        exactly what an untrained TF graph holds.
    init='he': uniform with var 2/fan_in for the hidden layers so activations stay O(1)
        through the 12-layer stack (makes the absolute 1e-4 parity tolerance meaningful).
    flow_gain scales Temporal/prediction so the softmax over the 64 window cells is not
        uniform and the soft-argmax flow / bilinear warp are actually exercised.
    """
    rng = np.random.default_rng(seed)
    W = {}
    for name, kind, shape in variable_specs():
        fan_in, fan_out = _fans(kind, shape)
        if init == 'glorot':
            lim = np.sqrt(6.0 / (fan_in + fan_out))
        elif init == 'he':
            lim = np.sqrt(6.0 / fan_in)
        else:
            raise ValueError(init)
        k = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        if name == 'Temporal/prediction':
            k *= np.float32(flow_gain)
        if name in ('ScoreNet/prediction', 'Temporal/uncertainty'):
            k *= np.float32(0.25)   # keep exp() heads in a sane range
        W[name + '/kernel'] = k
        nb = shape[2] if kind == 'deconv' else shape[-1]
        W[name + '/bias'] = (rng.uniform(-1, 1, size=(nb,)) * bias_scale).astype(np.float32)
    return W


def save_npz(path, W):
    np.savez(path, **W)


def load_npz(path):
    with np.load(path) as z:
        return {k: z[k].astype(np.float32) for k in z.files}


def from_network_load_dict(data_dict, scope):
    """Flatten `Network.load`'s {op_name: {param_name: array}} (network.py:60-75)."""
    out = {}
    for op_name, params in data_dict.items():
        for pname, arr in params.items():
            out['%s/%s/%s' % (scope, op_name, pname)] = np.asarray(arr, dtype=np.float32)
    return out


def num_params(W):
    return int(sum(v.size for v in W.values()))
