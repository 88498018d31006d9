"""Helpers for the -m gpu tests: device buffers via torch, calls via the ctypes C ABI."""
import ctypes as C

import numpy as np

from kfnet_amd import _lib
from kfnet_amd.graph import pack_conv_kernel, pack_conv_kernel_chunked, pack_deconv_kernel


def dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def sync():
    import torch
    torch.cuda.synchronize()


def run_conv(x, w, b, stride=1, relu=False, transposed=False, epilogue=0, config=0, ldx=None, ldy=None,
             x_off=0, y_off=0, f16=False, x3=False, x16=False, y16=False, k_step=0, weights_path=0):
    """x [N,H,W,Cin] np fp32; w TF layout; returns y np [N,Ho,Wo,Cout] computed by the HIP library.
    ldx/ldy > C exercise the strided-view paths (input/output living in wider buffers)."""
    import torch
    lib = _lib.load()
    n, h, wd, cin = x.shape
    if transposed:
        kh, kw, cout, _ = w.shape
        ho, wo = h * stride, wd * stride
        wp = pack_deconv_kernel(w)
    else:
        kh, kw, _, cout = w.shape
        ho, wo = -(-h // stride), -(-wd // stride)
        wp = pack_conv_kernel(w)
    ldx = ldx or cin
    ldy = ldy or cout
    # x16 / y16: the activation tensors hold IEEE halfs in memory (kfn_conv_desc.x_dtype / y_dtype)
    xb = np.full((n * h * wd, ldx), 7.0, dtype=np.float16 if x16 else np.float32)
    xb[:, x_off:x_off + cin] = x.reshape(-1, cin)
    xd = dev(xb)
    GUARD = 96   # rows behind the tensor: the last (partial) tile must not write past row M
    yd = torch.full((n * ho * wo + GUARD, ldy), -123.0, dtype=torch.float16 if y16 else torch.float32, device='cuda')
    xsz, ysz = (2 if x16 else 4), (2 if y16 else 4)
    if x3:
        m = wp * np.float32(1024.0)
        hi = m.astype(np.float16)
        wp_dev = np.stack([hi, (m - hi.astype(np.float32)).astype(np.float16)])
    else:
        wp_dev = wp.astype(np.float16) if f16 else wp
        if x16 or y16:      # fp16 activations: chunk-major weights
            wp_dev = pack_conv_kernel_chunked(w).astype(np.float16)
    wd_ = dev(wp_dev)
    bd = dev(b.astype(np.float32)) if b is not None else None
    d = _lib.ConvDesc(N=n, H=h, W=wd, Cin=cin, ldx=ldx, Cout=cout, cout_pad=wp.shape[0], ldy=ldy, kh=kh, kw=kw,
                      stride=stride, transposed=int(transposed), relu=int(relu), epilogue=epilogue, config=config,
                      operand_dtype=2 if x3 else int(f16), x_dtype=int(x16), y_dtype=int(y16), k_step=k_step, weights_path=weights_path)
    rc = lib.kfn_conv2d_nhwc(C.byref(d), xd.data_ptr() + xsz * x_off, wd_.data_ptr(),
                             bd.data_ptr() if bd is not None else None, yd.data_ptr() + ysz * y_off, stream())
    _lib.check(rc, 'kfn_conv2d_nhwc')
    sync()
    yh = yd.cpu().numpy().astype(np.float32)
    assert np.all(yh[n * ho * wo:] == -123.0), 'conv wrote past the last output row'
    yh = yh[:n * ho * wo]
    out = yh[:, y_off:y_off + cout].reshape(n, ho, wo, cout)
    # untouched columns must keep the sentinel
    mask = np.ones(ldy, dtype=bool)
    mask[y_off:y_off + cout] = False
    assert np.all(yh[:, mask] == -123.0), 'conv wrote outside its channel window'
    return out


def run_winograd(kind, x, wt, b, relu=False, form=0, ldx_pad=0, ldy_pad=8, config=0, x_layout='nhwc', y_layout='nhwc'):
    """One launch of a minimal-filtering entry point on x [N,H,W,Cin] fp32 with the TF-layout 3x3 kernel wt:
      kind 'wino'  kfn_conv2d_winograd        (two kernels + workspace; F(2x2,3x3))
           'fused' kfn_conv2d_winograd_fused  (form 0 = the library's routing, 1 = KFN_WINO_FORM_ONE_WAVE)
           'f43'   kfn_conv2d_winograd_f43    (form 2 = four waves, 3 = eight waves)
           's2'    kfn_conv2d_winograd_s2     (stride 2; form 0 = four waves, 4 = eight waves, 5 = the F(4,2) form)
    Input and output live in wider buffers (ldx = Cin + ldx_pad filled with 9.0 behind Cin, ldy = Cout + ldy_pad); guard
    rows behind the output and the columns behind Cout must come back untouched.  x_layout / y_layout 'c16': that tensor is handed
    over / comes back channel-blocked (KFN_LAYOUT_C16, per image [C/16][H][W][16]; dense, so its pad is dropped); the conversion
    from / to NHWC happens here on the host.  Returns y [N,Ho,Wo,Cout] np fp32."""
    import torch
    from kfnet_amd.graph import (pack_winograd_f43_kernel, pack_winograd_f43_kernel_b, pack_winograd_fused_kernel,
                                 pack_winograd_kernel, pack_winograd_s2_kernel, pack_winograd_s2_kernel_b, pack_winograd_s2_kernel_c)
    lib = _lib.load()
    n, h, w, ci = x.shape
    co = wt.shape[3]
    stride = 2 if kind == 's2' else 1
    ho, wo = -(-h // stride), -(-w // stride)
    if x_layout == 'c16':
        ldx_pad = 0
    if y_layout == 'c16':
        ldy_pad = 0
    ldx, ldy = ci + ldx_pad, co + ldy_pad
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=ci, ldx=ldx, Cout=co, cout_pad=-(-co // 32) * 32, ldy=ldy, kh=3, kw=3,
                      stride=stride, relu=int(relu), wino_form=form, config=config,
                      x_layout=_lib.LAYOUT_C16 if x_layout == 'c16' else 0, y_layout=_lib.LAYOUT_C16 if y_layout == 'c16' else 0)
    if x_layout == 'c16':
        xb = np.ascontiguousarray(x.reshape(n, h, w, ci // 16, 16).transpose(0, 3, 1, 2, 4)).reshape(n * h * w, ci)
    else:
        xb = np.full((n * h * w, ldx), 9.0, dtype=np.float32)
        xb[:, :ci] = x.reshape(-1, ci)
    GUARD = 64
    y = torch.full((n * ho * wo + GUARD, ldy), -5.0, device='cuda')
    dx = dev(xb)
    db = dev(np.concatenate([b, np.zeros((-co) % 4, np.float32)]).astype(np.float32)) if b is not None else None
    bp = db.data_ptr() if db is not None else None
    if kind == 'wino':
        nb = C.c_size_t()
        _lib.check(lib.kfn_winograd_workspace_bytes(C.byref(d), C.byref(nb)), 'ws')
        ws = torch.empty((nb.value // 4 + 64,), device='cuda')
        du = dev(pack_winograd_kernel(wt))
        _lib.check(lib.kfn_conv2d_winograd(C.byref(d), dx.data_ptr(), du.data_ptr(), bp, y.data_ptr(), ws.data_ptr(), 3,
                                           stream()), 'kfn_conv2d_winograd')
    elif kind == 'fused':
        du = dev(pack_winograd_fused_kernel(wt))
        _lib.check(lib.kfn_conv2d_winograd_fused(C.byref(d), dx.data_ptr(), du.data_ptr(), bp, y.data_ptr(), stream()),
                   'kfn_conv2d_winograd_fused')
    elif kind == 'f43':
        du = dev((pack_winograd_f43_kernel_b if form == 3 else pack_winograd_f43_kernel)(wt))
        _lib.check(lib.kfn_conv2d_winograd_f43(C.byref(d), dx.data_ptr(), du.data_ptr(), bp, y.data_ptr(), stream()),
                   'kfn_conv2d_winograd_f43')
    elif kind == 's2':
        du = dev({4: pack_winograd_s2_kernel_b, 5: pack_winograd_s2_kernel_c}.get(form, pack_winograd_s2_kernel)(wt))
        _lib.check(lib.kfn_conv2d_winograd_s2(C.byref(d), dx.data_ptr(), du.data_ptr(), bp, y.data_ptr(), stream()),
                   'kfn_conv2d_winograd_s2')
    else:
        raise ValueError(kind)
    sync()
    got = y.cpu().numpy()
    assert np.all(got[n * ho * wo:] == -5.0), '%s wrote past the last output pixel' % kind
    assert np.all(got[:, co:] == -5.0), '%s wrote outside its channel window' % kind
    if y_layout == 'c16':
        return np.ascontiguousarray(got[:n * ho * wo].reshape(n, co // 16, ho, wo, 16).transpose(0, 2, 3, 1, 4)).reshape(n, ho, wo, co)
    return got[:n * ho * wo, :co].reshape(n, ho, wo, co)
