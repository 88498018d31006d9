"""CPU ORACLE #2 (test infrastructure, NOT product code) -- torch-CPU fp32 restatement,
written independently of oracle/kfnet_oracle.py (different conv engine: oneDNN conv2d
in NCHW vs. numpy per-tap GEMMs) so that a mis-remembered TF semantic in one of them
shows up as a disagreement.  PARITY UNPINNED (see kfnet_oracle.py header).

It is also the "reference-faithful CPU restatement" that bench.py times as
`cpu_baseline` (kind "port"): `eval_step_reference_style` executes one eval.py step the
way the reference does it -- both towers on a 2-frame batch (KFNet/eval.py:41), 64
materialised feature shifts (KFNet/KFNet.py:348-357), unfused elementwise ops.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

MIN_UNCERTAINTY = 1e-5


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def conv_same(x, w, b, stride, relu):
    """x NCHW fp32; w TF HWIO ndarray/tensor.  TF-SAME (asymmetric) padding via F.pad
    (cnn_wrapper/network.py:116-135)."""
    w = w if torch.is_tensor(w) else _t(w)
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = _same_pad(x.shape[2], kh, stride)
    _, pl, pr = _same_pad(x.shape[3], kw, stride)
    if pt or pb or pl or pr:
        x = F.pad(x, (pl, pr, pt, pb))
    wt = w.permute(3, 2, 0, 1).contiguous()  # OIHW
    bb = None if b is None else (b if torch.is_tensor(b) else _t(b))
    y = F.conv2d(x, wt, bb, stride=stride)
    return F.relu(y) if relu else y


def deconv_same(x, w, b, stride, relu):
    """tf.layers.conv2d_transpose k3 s2 SAME (network.py:418-437):
    conv_transpose2d(padding=0) gives (n-1)*s+k rows; TF keeps rows [pad_before, pad_before+s*n).
    w is [kh,kw,Cout,Cin]; torch wants [Cin,Cout,kh,kw]."""
    w = w if torch.is_tensor(w) else _t(w)
    kh, kw = w.shape[0], w.shape[1]
    H, Wd = x.shape[2], x.shape[3]
    wt = w.permute(3, 2, 0, 1).contiguous()
    y = F.conv_transpose2d(x, wt, None, stride=stride)
    _, pt, _ = _same_pad(H * stride, kh, stride)
    _, pl, _ = _same_pad(Wd * stride, kw, stride)
    y = y[:, :, pt:pt + H * stride, pl:pl + Wd * stride]
    if y.shape[2] < H * stride or y.shape[3] < Wd * stride:
        y = F.pad(y, (0, Wd * stride - y.shape[3], 0, H * stride - y.shape[2]))
    if b is not None:
        y = y + (b if torch.is_tensor(b) else _t(b)).view(1, -1, 1, 1)
    return F.relu(y) if relu else y


def _kb(W, scope, name):
    return W['%s/%s/kernel' % (scope, name)], W.get('%s/%s/bias' % (scope, name))


SCOORD = [('conv1a', 1, True), ('conv1b', 1, True), ('conv2a', 2, True), ('conv2b', 1, True),
          ('conv3a', 2, True), ('conv3b', 1, True), ('conv4a', 2, True), ('conv4b', 1, True),
          ('conv5', 1, True), ('conv6', 1, True), ('conv7', 1, True), ('prediction', 1, False)]
FEAT = [('feat1', 1, True), ('feat2', 2, True), ('feat3', 1, True), ('feat4', 2, True),
        ('feat5', 1, True), ('feat6', 2, True), ('feat7', 1, False)]


def _pre(images):
    """images [B,H,W,3] (uint8 or float) -> NCHW fp32 preprocessed."""
    x = _t(images).to(torch.float32).permute(0, 3, 1, 2)
    return (x - 128.0) * 0.00625


@torch.no_grad()
def scoordnet(images, W, return_layers=False):
    x = _pre(images)
    L = {}
    for name, s, relu in SCOORD:
        x = conv_same(x, *_kb(W, 'ScoreNet', name), s, relu)
        if return_layers:
            L[name] = x.permute(0, 2, 3, 1).numpy()
    y = x.permute(0, 2, 3, 1)
    coord = y[..., 0:3].contiguous().numpy()
    unc = torch.exp(y[..., 3:4]).contiguous().numpy()
    if return_layers:
        return coord, unc, L
    return coord, unc


@torch.no_grad()
def oflow_feat(images, W):
    x = _pre(images)
    for name, s, relu in FEAT:
        x = conv_same(x, *_kb(W, 'Temporal', name), s, relu)
    y = x.permute(0, 2, 3, 1)
    ss = (y * y).sum(-1, keepdim=True)
    y = y * torch.rsqrt(torch.clamp(ss, min=1e-12))
    return y.contiguous().numpy()


@torch.no_grad()
def coord_volume(f1, f2, window=8):
    """64 materialised NEAREST/zero-fill shifts, like the reference (KFNet.py:343-359)."""
    f1 = _t(f1)
    f2 = _t(f2)
    _, h, w, C = f1.shape
    half = window // 2
    diffs = []
    offs = []
    for i in range(window):
        for j in range(window):
            dy, dx = i - half, j - half
            sh = torch.zeros_like(f1)
            ys0, ys1 = max(0, -dy), min(h, h - dy)
            xs0, xs1 = max(0, -dx), min(w, w - dx)
            if ys1 > ys0 and xs1 > xs0:
                sh[:, ys0:ys1, xs0:xs1, :] = f1[:, ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx, :]
            diffs.append(f2 - sh)
            offs.append((dx, dy))
    vol = torch.cat(diffs, dim=-1).reshape(-1, window, window, C)
    return vol.numpy(), np.array(offs, dtype=np.float32)


@torch.no_grad()
def oflownet(vol, W, chunk=1200):
    S = 'Temporal'
    vol = _t(vol)
    probs, sig = [], []
    for s0 in range(0, vol.shape[0], chunk):
        x = vol[s0:s0 + chunk].permute(0, 3, 1, 2).contiguous()
        c0 = conv_same(x, *_kb(W, S, 'conv0'), 1, True)
        c1a = conv_same(c0, *_kb(W, S, 'conv1a'), 2, True)
        c1b = conv_same(c1a, *_kb(W, S, 'conv1b'), 1, True)
        c2a = conv_same(c1b, *_kb(W, S, 'conv2a'), 2, True)
        c2b = conv_same(c2a, *_kb(W, S, 'conv2b'), 1, True)
        c3a = conv_same(c2b, *_kb(W, S, 'conv3a'), 2, True)
        c3b = conv_same(c3a, *_kb(W, S, 'conv3b'), 1, True)
        u2 = deconv_same(c3b, *_kb(W, S, 'upconv2'), 2, True)
        c4 = conv_same(torch.cat([u2, c2b], 1), *_kb(W, S, 'conv4'), 1, True)
        u1 = deconv_same(c4, *_kb(W, S, 'upconv1'), 2, True)
        c5 = conv_same(torch.cat([u1, c1b], 1), *_kb(W, S, 'conv5'), 1, True)
        u0 = deconv_same(c5, *_kb(W, S, 'upconv0'), 2, True)
        c6 = conv_same(torch.cat([u0, c0], 1), *_kb(W, S, 'conv6'), 1, True)
        pr = conv_same(c6, *_kb(W, S, 'prediction'), 1, False)
        n = x.shape[0]
        prob = torch.softmax(pr[:, 0].reshape(n, -1), dim=-1)
        feat = c3b.reshape(n, -1)
        k1, b1 = _kb(W, S, 'fc1')
        k2, b2 = _kb(W, S, 'fc2')
        k3, b3 = _kb(W, S, 'uncertainty')
        f1 = F.relu(feat @ _t(k1) + _t(b1))
        f2 = F.relu(f1 @ _t(k2) + _t(b2))
        u = f2 @ _t(k3) + _t(b3)
        probs.append(prob)
        sig.append(torch.exp(u) * 1e-2)
    return torch.cat(probs).numpy(), torch.cat(sig).numpy()


def bilinear_sampler(imgs, coords):
    """tools/util.py:36-93 in fp32 torch."""
    imgs = _t(imgs).float()
    coords = _t(coords).float()
    _, H, Wd, C = imgs.shape
    x = coords[..., 0:1]
    y = coords[..., 1:2]
    x0 = torch.floor(x); x1 = x0 + 1
    y0 = torch.floor(y); y1 = y0 + 1
    x0s = x0.clamp(0, Wd - 1); x1s = x1.clamp(0, Wd - 1)
    y0s = y0.clamp(0, H - 1); y1s = y1.clamp(0, H - 1)
    wx0 = x1s - x; wx1 = x - x0s
    wy0 = y1s - y; wy1 = y - y0s
    flat = imgs.reshape(-1, C)
    def g(xx, yy):
        return flat[(xx + yy * Wd).to(torch.int64)[..., 0]]
    out = (wx0 * wy0) * g(x0s, y0s) + (wx0 * wy1) * g(x0s, y1s)
    out = out + (wx1 * wy0) * g(x1s, y0s)
    out = out + (wx1 * wy1) * g(x1s, y1s)
    return out.numpy()


def process_model(prob, sigma_trans, offsets, last_coord, last_unc):
    _, h, w, _ = last_coord.shape
    flow = (_t(prob).float() @ _t(offsets).float()).reshape(1, h, w, 2).numpy()
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    pm = np.stack([xs, ys], -1)[None] + flow
    tx = bilinear_sampler(last_coord, pm)
    lu = bilinear_sampler(last_unc, pm)
    eps2 = np.float32(MIN_UNCERTAINTY) * np.float32(MIN_UNCERTAINTY)
    lv = np.maximum(lu * lu, eps2)
    st = sigma_trans.reshape(1, h, w, 1).astype(np.float32)
    tv = np.maximum(st * st, eps2)
    return tx, np.sqrt(tv + lv), flow


def build_kf_coord(lx, ls, zx, zs):
    lv = ls * ls
    mv = zs * zs
    K = lv / (lv + mv)
    om = np.maximum(np.float32(1.0) - K, np.float32(0))
    return om * lx + K * zx, np.sqrt(om * lv)


def apply_transform(coords, T):
    T = np.asarray(T, dtype=np.float32)
    return (coords @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


@torch.no_grad()
def eval_step_reference_style(img_pair, W, state_x, state_s, transform, reset):
    """One KFNet/eval.py step executed like the reference (batch-2 towers, 64 shifts).
    img_pair [2,H,W,3]; returns record [h,w,4], new state."""
    coord, unc = scoordnet(img_pair, W)          # batch 2, only [1] used (KFNet.py:470-474)
    feats = oflow_feat(img_pair, W)              # batch 2
    z, sz = coord[1:2], unc[1:2]
    vol, offs = coord_volume(feats[0:1], feats[1:2], 8)
    prob, st = oflownet(vol, W)
    if state_x is None:
        state_x, state_s = np.zeros_like(z), np.ones_like(sz)
    tx, ts, _ = process_model(prob, st, offs, state_x, state_s)
    kx, ks = build_kf_coord(tx, ts, z, sz)
    if reset:
        ox, os_, nx, ns = apply_transform(z, transform), sz, z, sz
    else:
        ox, os_, nx, ns = apply_transform(kx, transform), ks, kx, ks
    rec = np.concatenate([ox[0], 1.0 / os_[0]], -1).astype(np.float32)
    return rec, nx, ns


@torch.no_grad()
def eval_sequence(images, W, transform, reset_period=500, dedup=True):
    """fp32 eval loop.  dedup=False runs the reference-faithful batch-2 step."""
    T = images.shape[0]
    recs = []
    sx = ss = None
    f_prev = None
    for i in range(T):
        reset = (i % reset_period == 0)
        if not dedup:
            pair = np.stack([images[i + 1 if i == 0 and T > 1 else max(i - 1, 0)], images[i]])
            rec, sx, ss = eval_step_reference_style(pair, W, sx, ss, transform, reset)
            recs.append(rec)
            continue
        z, sz = scoordnet(images[i:i + 1], W)
        f = oflow_feat(images[i:i + 1], W)
        if reset:
            ox, os_ = apply_transform(z, transform), sz
            sx, ss = z, sz
        else:
            vol, offs = coord_volume(f_prev, f, 8)
            prob, st = oflownet(vol, W)
            tx, ts, _ = process_model(prob, st, offs, sx, ss)
            kx, ks = build_kf_coord(tx, ts, z, sz)
            ox, os_ = apply_transform(kx, transform), ks
            sx, ss = kx, ks
        f_prev = f
        recs.append(np.concatenate([ox[0], 1.0 / os_[0]], -1).astype(np.float32))
    return np.stack(recs)
