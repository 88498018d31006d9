"""CPU ORACLE (test infrastructure, NOT product code) -- numpy restatement of the
KFNet per-sequence prediction path of zlthinker/KFNet.

PARITY UNPINNED: the reference is Python-2 + TensorFlow-1.x (cnn_wrapper/network.py,
KFNet/KFNet.py); TensorFlow is not installable in this environment and the reference
ships no tests, golden vectors, weights or sample data.  This file
therefore restates the algorithm from the cited reference lines + documented TF-1.x op
semantics, and is pinned only by (1) the analytic known-answer tests
in tests/test_oracle_kat.py, (2) agreement with the independent torch-CPU restatement
in oracle/kfnet_oracle_torch.py, (3) the committed fixtures in tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

All tensors are NHWC.  `dtype` selects the arithmetic type: float64 = "gold",
float32 = same-precision restatement.
Weights are a flat dict keyed by the TF variable names of SURVEY.md App. B, e.g.
"ScoreNet/conv1a/kernel" [3,3,3,64] (HWIO), "Temporal/upconv2/kernel" [3,3,64,128]
([k,k,Cout,Cin], the tf.layers.conv2d_transpose layout).
"""
import numpy as np

MIN_UNCERTAINTY = 1e-5  # KFNet/KFNet.py:71


# --------------------------------------------------------------------------------------
# TF op semantics
# --------------------------------------------------------------------------------------
def same_pad(in_size, k, stride):
    """TF 'SAME' padding: returns (out, pad_before, pad_after)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv2d_same(x, w, b=None, stride=1, relu=False):
    """tf.layers.conv2d(padding='SAME') as used by Network.conv
    (cnn_wrapper/network.py:116-135).  x [B,H,W,Cin], w [kh,kw,Cin,Cout] (HWIO),
    cross-correlation, y[o] = sum_k x[o*s + k - pad_before] * w[k]."""
    B, H, W, Cin = x.shape
    kh, kw, wcin, Cout = w.shape
    assert wcin == Cin, (wcin, Cin)
    Ho, pt, pb = same_pad(H, kh, stride)
    Wo, pl, pr = same_pad(W, kw, stride)
    xp = np.zeros((B, H + pt + pb, W + pl + pr, Cin), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    y = np.zeros((B, Ho, Wo, Cout), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            y += patch.reshape(-1, Cin).dot(w[i, j].astype(x.dtype)).reshape(B, Ho, Wo, Cout)
    if b is not None:
        y += b.astype(x.dtype)
    if relu:
        y = np.maximum(y, 0)
    return y


def conv2d_transpose_same(x, w, b=None, stride=2, relu=False):
    """tf.layers.conv2d_transpose(padding='SAME') as used by Network.deconv
    (cnn_wrapper/network.py:418-437), SURVEY App. A2.  w [kh,kw,Cout,Cin].
    Defined as the input-gradient of the SAME forward conv (out s*n -> n):
    y[s*o + k - pad_before] += x[o] * w[k], rows outside [0, s*n) dropped."""
    B, H, W, Cin = x.shape
    kh, kw, Cout, wcin = w.shape
    assert wcin == Cin, (wcin, Cin)
    Ho, Wo = H * stride, W * stride
    # padding of the forward conv that maps Ho -> H
    _, pt, _ = same_pad(Ho, kh, stride)
    _, pl, _ = same_pad(Wo, kw, stride)
    full = np.zeros((B, (H - 1) * stride + kh, (W - 1) * stride + kw, Cout), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            contrib = x.reshape(-1, Cin).dot(w[i, j].astype(x.dtype).T).reshape(B, H, W, Cout)
            full[:, i:i + (H - 1) * stride + 1:stride, j:j + (W - 1) * stride + 1:stride, :] += contrib
    # crop: output index p = s*o + k - pad_before
    yfull = np.zeros((B, Ho, Wo, Cout), dtype=x.dtype)
    hh = min(Ho, full.shape[1] - pt)
    ww = min(Wo, full.shape[2] - pl)
    yfull[:, :hh, :ww, :] = full[:, pt:pt + hh, pl:pl + ww, :]
    y = yfull
    if b is not None:
        y = y + b.astype(x.dtype)
    if relu:
        y = np.maximum(y, 0)
    return y


def dense(x, w, b=None, relu=False):
    """tf.layers.dense: kernel [in,out] (cnn_wrapper/OFlowNet.py:50-55)."""
    y = x.dot(w.astype(x.dtype))
    if b is not None:
        y = y + b.astype(x.dtype)
    if relu:
        y = np.maximum(y, 0)
    return y


def softmax(x):
    """tf.nn.softmax(axis=-1) (cnn_wrapper/OFlowNet.py:47)."""
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def l2_normalize(x, eps=1e-12):
    """tf.nn.l2_normalize(axis=-1): x * rsqrt(max(sum x^2, eps)) (KFNet/KFNet.py:340)."""
    ss = (x * x).sum(axis=-1, keepdims=True)
    return x / np.sqrt(np.maximum(ss, x.dtype.type(eps)))


def preprocess(images, dtype):
    """(x - 128) * 0.00625 (cnn_wrapper/SCoordNet.py:34-37, KFNet/KFNet.py:317)."""
    x = images.astype(dtype)
    return (x - dtype(128.0)) * dtype(0.00625)


# --------------------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------------------
SCOORD_LAYERS = [  # (name, k, Cout, stride, relu)  cnn_wrapper/SCoordNet.py:21-32
    ('conv1a', 3, 64, 1, True), ('conv1b', 3, 64, 1, True),
    ('conv2a', 3, 256, 2, True), ('conv2b', 3, 256, 1, True),
    ('conv3a', 3, 512, 2, True), ('conv3b', 3, 512, 1, True),
    ('conv4a', 3, 1024, 2, True), ('conv4b', 3, 1024, 1, True),
    ('conv5', 3, 512, 1, True), ('conv6', 3, 256, 1, True),
    ('conv7', 1, 128, 1, True), ('prediction', 1, 4, 1, False),
]

FEAT_LAYERS = [  # KFNet/KFNet.py:318-338
    ('feat1', 3, 16, 1, True), ('feat2', 3, 32, 2, True), ('feat3', 3, 32, 1, True),
    ('feat4', 3, 64, 2, True), ('feat5', 3, 64, 1, True), ('feat6', 3, 128, 2, True),
    ('feat7', 3, 32, 1, False),
]


def _kb(W, scope, name):
    return W['%s/%s/kernel' % (scope, name)], W.get('%s/%s/bias' % (scope, name))


def scoordnet(images, W, dtype=np.float64, return_layers=False):
    """SCoordNet forward + GetOutput (cnn_wrapper/SCoordNet.py:18-44).
    images [B,H,W,3] in 0..255 -> coord [B,h,w,3], uncertainty [B,h,w,1] = exp(ch3)."""
    x = preprocess(images, dtype)
    layers = {}
    for name, k, cout, s, relu in SCOORD_LAYERS:
        w, b = _kb(W, 'ScoreNet', name)
        x = conv2d_same(x, w, b, s, relu)
        layers[name] = x
    coord = x[..., 0:3]
    unc = np.exp(x[..., 3:4])
    if return_layers:
        return coord, unc, layers
    return coord, unc


def oflow_feat(images, W, dtype=np.float64, return_layers=False):
    """KFNet.BuildOFlowFeat (KFNet/KFNet.py:315-341): 7 convs + l2_normalize."""
    x = preprocess(images, dtype)
    layers = {}
    for name, k, cout, s, relu in FEAT_LAYERS:
        w, b = _kb(W, 'Temporal', name)
        x = conv2d_same(x, w, b, s, relu)
        layers[name] = x
    x = l2_normalize(x)
    if return_layers:
        return x, layers
    return x


def coord_volume(f1, f2, window=8):
    """KFNet.BuildCoordVolume + reshape (KFNet/KFNet.py:343-359, :372).
    f1,f2 [1,h,w,C] -> V [h*w, window, window, C] with
    V[p,i,j,c] = f2[y,x,c] - f1[y+i-half, x+j-half, c] (0 outside; tf.contrib.image.translate
    NEAREST zero fill) and offsets [window^2, 2] = (j-half, i-half) = (x,y)."""
    _, h, w, C = f1.shape
    half = window // 2
    f1p = np.zeros((h + window, w + window, C), dtype=f1.dtype)
    f1p[half:half + h, half:half + w, :] = f1[0]
    V = np.zeros((h, w, window, window, C), dtype=f1.dtype)
    offsets = []
    for i in range(window):
        for j in range(window):
            V[:, :, i, j, :] = f2[0] - f1p[i:i + h, j:j + w, :]
            offsets.append((j - half, i - half))
    return V.reshape(h * w, window, window, C), np.array(offsets, dtype=f1.dtype)


def oflownet(vol, W, dtype=np.float64, return_layers=False):
    """OFlowNet.setup + GetOutput (cnn_wrapper/OFlowNet.py:17-57).
    vol [N,8,8,32] -> prob [N,64], sigma_trans [N,1]."""
    S = 'Temporal'
    L = {}
    x = vol.astype(dtype)
    L['conv0'] = conv2d_same(x, *_kb(W, S, 'conv0'), 1, True)
    L['conv1a'] = conv2d_same(L['conv0'], *_kb(W, S, 'conv1a'), 2, True)
    L['conv1b'] = conv2d_same(L['conv1a'], *_kb(W, S, 'conv1b'), 1, True)
    L['conv2a'] = conv2d_same(L['conv1b'], *_kb(W, S, 'conv2a'), 2, True)
    L['conv2b'] = conv2d_same(L['conv2a'], *_kb(W, S, 'conv2b'), 1, True)
    L['conv3a'] = conv2d_same(L['conv2b'], *_kb(W, S, 'conv3a'), 2, True)
    L['conv3b'] = conv2d_same(L['conv3a'], *_kb(W, S, 'conv3b'), 1, True)
    L['upconv2'] = conv2d_transpose_same(L['conv3b'], *_kb(W, S, 'upconv2'), 2, True)
    L['concat2'] = np.concatenate([L['upconv2'], L['conv2b']], axis=-1)
    L['conv4'] = conv2d_same(L['concat2'], *_kb(W, S, 'conv4'), 1, True)
    L['upconv1'] = conv2d_transpose_same(L['conv4'], *_kb(W, S, 'upconv1'), 2, True)
    L['concat1'] = np.concatenate([L['upconv1'], L['conv1b']], axis=-1)
    L['conv5'] = conv2d_same(L['concat1'], *_kb(W, S, 'conv5'), 1, True)
    L['upconv0'] = conv2d_transpose_same(L['conv5'], *_kb(W, S, 'upconv0'), 2, True)
    L['concat0'] = np.concatenate([L['upconv0'], L['conv0']], axis=-1)
    L['conv6'] = conv2d_same(L['concat0'], *_kb(W, S, 'conv6'), 1, True)
    L['prediction'] = conv2d_same(L['conv6'], *_kb(W, S, 'prediction'), 1, False)
    N = vol.shape[0]
    logits = L['prediction'][..., 0].reshape(N, -1)
    prob = softmax(logits)
    feat = L['conv3b'].reshape(N, -1)
    fc1 = dense(feat, *_kb(W, S, 'fc1'), relu=True)
    fc2 = dense(fc1, *_kb(W, S, 'fc2'), relu=True)
    unc = dense(fc2, *_kb(W, S, 'uncertainty'), relu=False)
    sigma_trans = np.exp(unc) * dtype(1e-2)
    if return_layers:
        L['logits'] = logits
        return prob, sigma_trans, L
    return prob, sigma_trans


# --------------------------------------------------------------------------------------
# process model tail, Kalman fuse, NIS, transform
# --------------------------------------------------------------------------------------
def get_pixel_map(h, w, dtype=np.float64):
    """KFNet/util.py:42-63: map[y,x] = (x, y)."""
    xs, ys = np.meshgrid(np.arange(w, dtype=dtype), np.arange(h, dtype=dtype))
    return np.stack([xs, ys], axis=-1)[None]


def bilinear_sampler(imgs, coords):
    """tools/util.py:3-94: clamped corner indices AND weights computed
    from the clamped corners; add_n order w00*im00 + w01*im01 + w10*im10 + w11*im11.
    imgs [1,H,W,C], coords [1,h,w,2] (x,y) -> [1,h,w,C]."""
    dt = imgs.dtype
    _, H, Wd, C = imgs.shape
    x = coords[..., 0:1].astype(dt)
    y = coords[..., 1:2].astype(dt)
    x0 = np.floor(x)
    x1 = x0 + 1
    y0 = np.floor(y)
    y1 = y0 + 1
    xmax = dt.type(Wd - 1)
    ymax = dt.type(H - 1)
    x0s = np.clip(x0, 0, xmax)
    x1s = np.clip(x1, 0, xmax)
    y0s = np.clip(y0, 0, ymax)
    y1s = np.clip(y1, 0, ymax)
    wx0 = x1s - x
    wx1 = x - x0s
    wy0 = y1s - y
    wy1 = y - y0s
    flat = imgs.reshape(-1, C)
    def g(xx, yy):
        idx = (xx + yy * dt.type(Wd)).astype(np.int32)[..., 0]
        return flat[idx]
    im00 = g(x0s, y0s)
    im01 = g(x0s, y1s)
    im10 = g(x1s, y0s)
    im11 = g(x1s, y1s)
    return ((wx0 * wy0) * im00 + (wx0 * wy1) * im01) + (wx1 * wy0) * im10 + (wx1 * wy1) * im11


def soft_argmax_flow(prob, offsets):
    """flow = prob[N,1,64] @ offsets[N,64,2] (KFNet/KFNet.py:381-385)."""
    return prob.dot(offsets.astype(prob.dtype))


def process_model(prob, sigma_trans, offsets, last_coord, last_unc):
    """KFNet.BuildOFlowNet tail (KFNet/KFNet.py:381-403).
    prob [h*w,64], sigma_trans [h*w,1], last_coord [1,h,w,3], last_unc [1,h,w,1]
    -> temp_coord [1,h,w,3], temp_unc [1,h,w,1], flow [1,h,w,2]."""
    dt = last_coord.dtype
    _, h, w, _ = last_coord.shape
    flow = soft_argmax_flow(prob.astype(dt), offsets).reshape(1, h, w, 2)
    pixel_map = get_pixel_map(h, w, dt) + flow
    temp_coord = bilinear_sampler(last_coord, pixel_map)
    last_u = bilinear_sampler(last_unc, pixel_map)
    eps2 = dt.type(MIN_UNCERTAINTY) * dt.type(MIN_UNCERTAINTY)
    last_var = np.maximum(last_u * last_u, eps2)
    st = sigma_trans.astype(dt).reshape(1, h, w, 1)
    trans_var = np.maximum(st * st, eps2)
    temp_unc = np.sqrt(trans_var + last_var)
    return temp_coord, temp_unc, flow


def build_kf_coord(last_coord, last_unc, measure_coord, measure_unc):
    """KFNet.BuildKFCoord (KFNet/KFNet.py:148-162)."""
    dt = last_coord.dtype
    last_var = last_unc * last_unc
    meas_var = measure_unc * measure_unc
    K = last_var / (last_var + meas_var)
    one_minus = np.maximum(dt.type(1.0) - K, 0)
    kf_coord = one_minus * last_coord + K * measure_coord
    kf_var = one_minus * last_var
    return kf_coord, np.sqrt(kf_var)


def get_kf_coord2(temp_coord, temp_unc, measure_coord, measure_unc):
    """KFNet.GetKFCoord2's fusion step (KFNet/KFNet.py:487-502): the symmetric-form posterior variance
    (1-K)^2 P^- + K^2 R; the mean as in BuildKFCoord."""
    dt = temp_coord.dtype
    meas_var = measure_unc * measure_unc
    temp_var = temp_unc * temp_unc
    K = temp_var / (temp_var + meas_var)
    kf_coord = np.maximum(dt.type(1.0) - K, 0) * temp_coord + K * measure_coord
    kf_var = (dt.type(1.0) - K) * (dt.type(1.0) - K) * temp_var + (K * K) * meas_var
    return kf_coord, np.sqrt(kf_var)


def get_nis(measure_coord, measure_unc, temp_coord, temp_unc):
    """KFNet.GetNIS (KFNet/KFNet.py:164-184).  Note inno_variance = square(sqrt(.))."""
    inno = measure_coord - temp_coord
    inno_unc = np.sqrt(temp_unc * temp_unc + measure_unc * measure_unc)
    return (inno * inno) / (inno_unc * inno_unc)


def apply_transform(coords, T):
    """KFNet/util.py:12-40: x' = (T @ [x;1])[0:3], no perspective divide."""
    dt = coords.dtype
    T = T.astype(dt)
    return coords.dot(T[:3, :3].T) + T[:3, 3]


def get_transform(mat4):
    """KFNet/train.py:49-58: transform = inv(loadtxt(transform.txt)) in float32."""
    return np.linalg.inv(np.asarray(mat4, dtype=np.float32))


# --------------------------------------------------------------------------------------
# eval.py loop
# --------------------------------------------------------------------------------------
def frame_stage(image, W, dtype):
    """Everything that depends on ONE image only: measurement + flow features."""
    z, sz = scoordnet(image[None], W, dtype)
    f = oflow_feat(image[None], W, dtype)
    return z, sz, f


def pair_stage(f1, f2, W, dtype):
    """Everything that depends on the image PAIR only: prob, sigma_trans, offsets."""
    vol, offsets = coord_volume(f1, f2, 8)
    prob, st = oflownet(vol, W, dtype)
    return prob, st, offsets


def eval_sequence(images, W, transform, reset_period=500, nis_gate=False,
                  dtype=np.float64, return_debug=False):
    """KFNet/eval.py:77-126 loop, de-duplicated form (each tower once per frame; results
    identical to the reference's 2-frame batches because no layer couples batch
    elements).  Schedule per KFNet/train.py:67-71: step 0 = pair (1,0) whose
    'frame 2' is frame 0; step k>=1 = pair (k-1,k).
    images [T,H,W,3] uint8; transform 4x4 (already inverted, see get_transform).
    Returns records [T,h,w,4] float32 = concat(T.x_KF, 1/sigma_KF) (eval.py:123-126)."""
    T = images.shape[0]
    records = []
    debug = []
    state_x = state_s = None
    f_prev = None
    for i in range(T):
        z, sz, f = frame_stage(images[i], W, dtype)
        d = {'z': z, 'sz': sz, 'feat': f}
        if i % reset_period == 0:
            # eval.py:94-101 state := measurement, outputs := measurement
            out_x, out_s = apply_transform(z, transform), sz
            state_x, state_s = z, sz
        else:
            prob, st, offsets = pair_stage(f_prev, f, W, dtype)
            tx, ts, flow = process_model(prob, st, offsets, state_x, state_s)
            kx, ks = build_kf_coord(tx, ts, z, sz)
            out_x, out_s = apply_transform(kx, transform), ks
            d.update({'prob': prob, 'sigma_trans': st, 'flow': flow, 'temp_x': tx, 'temp_s': ts,
                      'kf_x': kx, 'kf_s': ks})
            nis = get_nis(z, sz, tx, ts)
            d['nis'] = nis
            if nis_gate:  # eval.py:87-92: output only, state stays the KF estimate
                mask = (nis.sum(axis=-1, keepdims=True) > 7.815)
                out_x = np.where(mask, apply_transform(z, transform), out_x)
            state_x, state_s = kx, ks  # eval.py:103-104 (raw, untransformed, ungated)
        f_prev = f
        rec = np.concatenate([out_x[0], 1.0 / out_s[0]], axis=-1).astype(np.float32)
        records.append(rec)
        debug.append(d)
    records = np.stack(records)
    if return_debug:
        return records, debug
    return records
