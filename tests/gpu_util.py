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
