"""Deferred op graph that stands in for the TF-1.x graph + session of the reference.

The reference's `Network` layer methods add TF ops to a global graph that `sess.run`
executes (cnn_wrapper/network.py:8-31, KFNet/eval.py:78-83).  Here the same layer
methods append launch records for libkfnet_hip.so entry points to a `Graph`; `Graph.run`
replays them on a HIP stream.  Tensors are NHWC fp32 device buffers with an explicit
pixel stride so that `concat` is a re-binding of producer outputs, not a copy.

PyTorch is used only as the device allocator / stream provider.  Shape inference and
graph construction work without a GPU (device=None); running does not.
"""
import contextlib
import ctypes as C

import numpy as np

from . import _lib

_scope_stack = []


@contextlib.contextmanager
def variable_scope(name):
    """tf.variable_scope analogue (KFNet/KFNet.py:304,316,374): prefixes variable names."""
    _scope_stack.append(name)
    try:
        yield
    finally:
        _scope_stack.pop()


def current_scope():
    return '/'.join(_scope_stack)


ELEM_BYTES = {'f32': 4, 'f16': 2, 'u8': 1}


class _Shape(list):
    def as_list(self):
        return list(self)


class Storage(object):
    """A flat device buffer of `numel` elements of `dtype` ('f32' | 'f16' | 'u8')."""

    def __init__(self, numel, dtype='f32'):
        self.numel = int(numel)
        self.dtype = dtype
        self.buf = None  # torch tensor once allocated

    def allocate(self, device):
        import torch
        if self.buf is None:
            dt = {'f32': torch.float32, 'f16': torch.float16, 'u8': torch.uint8}[self.dtype]
            self.buf = torch.zeros(self.numel, dtype=dt, device=device)
        return self.buf

    @property
    def ptr(self):
        if self.buf is None:
            raise _lib.KfnError('graph buffers are not allocated: call Graph.finalize(device) '
                                '(needs a GPU; kfnet_amd has no CPU execution path)')
        return self.buf.data_ptr()


class Tensor(object):
    """NHWC tensor handle: a channel window [ch_off, ch_off+C) of a buffer whose pixel
    stride is `ld` elements.  `base` makes this a view that follows its parent when the
    parent is re-bound into a concat buffer.

    `layout`: 'nhwc' (every tensor a caller sees) or 'c16' -- per image [C/16][H][W][16] (include/kfnet_hip.h,
    KFN_LAYOUT_C16), given by Graph.assign_layouts to dense intermediate tensors whose producer and consumers are all
    Winograd launches that take it.  numpy() / upload() speak NHWC for both; channel views and concat windows of a c16
    tensor do not exist (they raise)."""

    def __init__(self, graph, shape, dtype='f32', name=None, base=None, rel_off=0, rel_batch=0):
        self.graph = graph
        self.shape = tuple(int(s) for s in shape)
        self.dtype = dtype
        self.name = name
        self.base = base
        self.rel_off = rel_off      # channel offset inside the parent
        self.rel_batch = rel_batch  # batch (outermost axis) offset inside the parent
        self.external = False
        if base is None:
            n, h, w, c = self.shape
            self.storage = Storage(n * h * w * c, dtype)
            self._ld = c
            self._off = 0
            self._slot = 0      # element offset of the tensor inside its buffer (slide(): a window that moves per batch)
            self._layout = 'nhwc'
            graph.storages.append(self.storage)

    # -- TF-flavoured introspection ----------------------------------------------------
    def get_shape(self):
        return _Shape(self.shape)

    @property
    def pixels(self):
        return self.shape[0] * self.shape[1] * self.shape[2]

    @property
    def C(self):
        return self.shape[3]

    @property
    def ld(self):
        return self.base.ld if self.base is not None else self._ld

    @property
    def layout(self):
        return self.base.layout if self.base is not None else self._layout

    def set_layout(self, layout):
        """Root tensors only, before any data is in the buffer (Graph.assign_layouts)."""
        if layout not in ('nhwc', 'c16'):
            raise ValueError('unknown layout %r' % (layout,))
        if layout == 'c16' and not (self.is_whole() and self.shape[3] % 16 == 0 and self.dtype == 'f32' and not self.external):
            raise ValueError('tensor %r cannot be channel-blocked: it must be a dense fp32 tensor the graph owns with C %% 16 == 0'
                             % self.name)
        self._layout = layout

    @property
    def ch_off(self):
        return (self.base.ch_off + self.rel_off) if self.base is not None else self._off

    @property
    def root_storage(self):
        return self.base.root_storage if self.base is not None else self.storage

    @property
    def elem_off(self):
        """Element offset of this view's batch window from the start of the root buffer."""
        if self.base is None:
            return self._slot
        _, h, w, _ = self.base.shape
        return self.base.elem_off + self.rel_batch * h * w * self.base.ld

    @property
    def ptr(self):
        return self.root_storage.ptr + (self.elem_off + self.ch_off) * ELEM_BYTES[self.dtype]

    def is_whole(self):
        return self.base is None and self._off == 0 and self._ld == self.shape[3]

    def rebind(self, storage, ch_off, ld):
        """Move this tensor's data into a window of a wider buffer (concat)."""
        assert self.base is None
        if self._layout != 'nhwc':
            raise ValueError('tensor %r is channel-blocked: it cannot become a window of a concat buffer' % self.name)
        if self.storage in self.graph.storages:
            self.graph.storages.remove(self.storage)
        self.storage = storage
        self._off = ch_off
        self._ld = ld

    def slide(self, elems):
        """Move this root tensor to element offset `elems` of its buffer: the launches that write / read it and every
        view of it follow at their next launch (pointers are taken at launch time).  The engine slides a [B,...] output
        along a [T,...] chunk buffer so that a batch's results land where the scan reads them (no copy)."""
        assert self.base is None
        n, h, w, _ = self.shape
        if elems < 0 or elems + n * h * w * self._ld > self.storage.numel:
            raise ValueError('slide(%d) leaves the buffer of tensor %r' % (elems, self.name))
        self._slot = int(elems)

    def channels(self, start, count, name=None):
        """tf.slice on the channel axis as a zero-copy view."""
        n, h, w, c = self.shape
        assert 0 <= start and start + count <= c
        if self.layout != 'nhwc':
            raise ValueError('tensor %r is channel-blocked: no channel views' % self.name)
        return Tensor(self.graph, (n, h, w, count), self.dtype, name, base=self, rel_off=start)

    def batch(self, start, count, name=None):
        """Zero-copy window [start, start+count) of the batch axis."""
        n, h, w, c = self.shape
        assert 0 <= start and start + count <= n
        return Tensor(self.graph, (count, h, w, c), self.dtype, name, base=self, rel_batch=start)

    # -- host <-> device ---------------------------------------------------------------
    def numpy(self):
        import torch
        n, h, w, c = self.shape
        buf = self.root_storage.buf
        if buf is None:
            raise _lib.KfnError('tensor %r is not allocated' % self.name)
        torch.cuda.synchronize()
        flat = buf.cpu().numpy()
        if self.layout == 'c16':        # per image [C/16][H][W][16] -> NHWC
            blk = flat[self.elem_off:self.elem_off + n * h * w * c].reshape(n, c // 16, h, w, 16)
            return np.ascontiguousarray(blk.transpose(0, 2, 3, 1, 4)).reshape(n, h, w, c)
        off, ld = self.ch_off + self.elem_off, self.ld
        idx = off + np.arange(n * h * w)[:, None] * ld + np.arange(c)[None, :]
        return flat[idx].reshape(n, h, w, c)

    def upload(self, arr):
        import torch
        n, h, w, c = self.shape
        arr = np.ascontiguousarray(arr).reshape(n, h, w, c)
        want = {'f32': np.float32, 'f16': np.float16, 'u8': np.uint8}[self.dtype]
        if arr.dtype != want:
            raise TypeError('tensor %r wants %s, got %s' % (self.name, want, arr.dtype))
        assert self.ld == c and self.ch_off == 0, 'upload needs a channel-dense tensor'
        if self.layout == 'c16':
            arr = np.ascontiguousarray(arr.reshape(n, h, w, c // 16, 16).transpose(0, 3, 1, 2, 4))
        dst = self.root_storage.buf[self.elem_off:self.elem_off + arr.size]
        dst.copy_(torch.from_numpy(arr.reshape(-1)), non_blocking=False)


class Param(object):
    """A weight variable: TF name + logical shape; `pack` turns the TF-layout ndarray
    into the device layout the kernels read."""

    def __init__(self, graph, name, shape, pack, source=None):
        self.name = name
        self.source = source or name   # TF variable it is filled from (derived layouts share one)
        self.shape = tuple(shape)
        self.pack = pack
        self.storage = None
        self.packed_shape = None
        graph.params[name] = self

    @property
    def ptr(self):
        if self.storage is None:
            raise _lib.KfnError('weights not loaded: variable %s (call Graph.load_weights)' % self.name)
        return self.storage.data_ptr()


# ---------------------------------------------------------------------------------------
# ops
# ---------------------------------------------------------------------------------------
def _scaled(n, graph):
    """Leading (batch) extent actually launched: graphs are built for B frames but a
    partial batch of nb <= B frames only runs its share (Graph.active = (nb, B))."""
    nb, B = graph.active
    return n if nb == B else (n * nb) // B


class Op(object):
    name = '?'

    def launch(self, lib, stream):
        raise NotImplementedError


def pack_conv_kernel(w):
    """TF HWIO [kh,kw,Cin,Cout] -> [cout_pad][kh*kw*Cin] (K contiguous), zero rows pad."""
    kh, kw, ci, co = w.shape
    cp = -(-co // 32) * 32
    out = np.zeros((cp, kh * kw * ci), dtype=np.float32)
    out[:co] = np.transpose(w, (3, 0, 1, 2)).reshape(co, -1)
    return out


def pack_conv_kernel_chunked(w, chunk=32):
    """TF HWIO [kh,kw,Cin,Cout] -> [K/chunk][cout_pad][chunk] with K = (kh,kw,ci) order: the layout of the
    fp16-activation convolution kernels (conv_mfma_kernel PREC 4-6, csrc/kfn_conv.hip), whose B tile of a stage is
    then one contiguous run -- every fetched cache line is fully used."""
    m = pack_conv_kernel(w)                            # [cout_pad][K]
    cp, K = m.shape
    assert K % chunk == 0
    return np.ascontiguousarray(m.reshape(cp, K // chunk, chunk).transpose(1, 0, 2))


_WINO_G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)


def pack_window_fc_kernel(w):
    """TF HWIO [3,3,Cin,Cout] -> the same 3x3 SAME stride-1 convolution on a 2x2 image written as ONE dense matrix,
    in the 1x1-kernel layout of pack_conv_kernel: row (q, o) x column (p, c) = w[py-qy+1, px-qx+1, c, o] with
    p = 2 py + px the input pixel and q = 2 qy + qx the output pixel.  On a 2x2 image every input pixel reaches
    every output pixel, so the matrix is dense: 16 Cin Cout products per window instead of the 36 Cin Cout the
    nine-tap form spends (20 of them on zero padding)."""
    w = np.asarray(w, np.float32)
    kh, kw, ci, co = w.shape
    assert kh == 3 and kw == 3
    m = np.zeros((4 * co, 4 * ci), np.float32)
    for q in range(4):
        for p in range(4):
            m[q * co:(q + 1) * co, p * ci:(p + 1) * ci] = w[(p >> 1) - (q >> 1) + 1, (p & 1) - (q & 1) + 1].T
    cp = -(-(4 * co) // 32) * 32
    out = np.zeros((cp, 4 * ci), np.float32)
    out[:4 * co] = m
    return out


def pack_bias_x4(b):
    """Bias of a layer run as a window matrix (pack_window_fc_kernel): one copy per output pixel."""
    return np.tile(pack_bias(b), 4)


def pack_winograd_kernel(w):
    """TF HWIO [3,3,Cin,Cout] -> U [16][cout_pad][Cin] with U[4*xi+nu] = (G g G^T)[xi][nu]
    (Winograd F(2x2,3x3) weight transform, evaluated in fp64, rounded once to fp32)."""
    kh, kw, ci, co = w.shape
    assert kh == 3 and kw == 3
    cp = -(-co // 32) * 32
    U = np.einsum('ai,ijco,bj->abco', _WINO_G, w.astype(np.float64), _WINO_G)  # [4,4,ci,co]
    out = np.zeros((16, cp, ci), dtype=np.float32)
    out[:, :co, :] = np.transpose(U.reshape(16, ci, co), (0, 2, 1)).astype(np.float32)
    return out


def pack_winograd_fused_kernel(w):
    """TF HWIO [3,3,Cin,Cout] -> U2 [Cin/8][16][cout_pad][8] for kfn_conv2d_winograd_fused: the same
    (G g G^T)[xi][nu] as pack_winograd_kernel, laid out so that the 32-channel fragment of one
    (8-channel k-chunk, position 4*xi+nu) is one contiguous 1 KiB run."""
    u = pack_winograd_kernel(w)                       # [16][cout_pad][Cin]
    g, cp, ci = u.shape
    assert ci % 8 == 0
    return np.ascontiguousarray(u.reshape(g, cp, ci // 8, 8).transpose(2, 0, 1, 3))


# Winograd F(4x4,3x3), interpolation points {0, +-1, +-2} (csrc/kfn_wino4.hip carries the same three matrices)
_WINO4_G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                     [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
_WINO4_BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                      [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
_WINO4_AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def pack_winograd_f43_kernel(w):
    """TF HWIO [3,3,Cin,Cout] -> U4 [Cin/8][36][cout_pad][8] for kfn_conv2d_winograd_f43: U[6*xi+nu] = (G g G^T)[xi][nu]
    of Winograd F(4x4,3x3), evaluated in fp64 and rounded once to fp32, laid out so that the 32-channel fragment of one
    (8-channel k-chunk, position) is one contiguous 1 KiB run."""
    kh, kw, ci, co = w.shape
    assert kh == 3 and kw == 3 and ci % 8 == 0
    cp = -(-co // 32) * 32
    U = np.einsum('ai,ijco,bj->abco', _WINO4_G, np.asarray(w, np.float64), _WINO4_G)   # [6,6,ci,co]
    u = np.zeros((36, cp, ci), dtype=np.float32)
    u[:, :co, :] = np.transpose(U.reshape(36, ci, co), (0, 2, 1)).astype(np.float32)
    return np.ascontiguousarray(u.reshape(36, cp, ci // 8, 8).transpose(2, 0, 1, 3))


def pack_winograd_f43_kernel_b(w):
    """The same U for the EIGHT-WAVE form of kfn_conv2d_winograd_f43 (kfn_conv_desc.wino_form = KFN_WINO_FORM_F43_EIGHT_WAVE,
    wino4b_kernel): U4b [Cin/8][18 position pairs][cout_pad][4 k][2 positions][2 k-steps] -- lane (channel n, k) of a
    16x16x4 B operand reads 16 contiguous bytes per (chunk, pair): positions 2q and 2q + 1, input channels 2k and 2k + 1."""
    u = pack_winograd_f43_kernel(w)                     # [Cin/8][36][cp][8]
    nc, _, cp, _ = u.shape
    v = u.reshape(nc, 18, 2, cp, 4, 2)                  # [chunk][pair][pp][n][k][s]
    return np.ascontiguousarray(v.transpose(0, 1, 3, 4, 2, 5).reshape(nc, 18, cp, 16))


def pack_winograd_s2_kernel(w):
    """TF HWIO [3,3,Cin,Cout] -> the 16 weight fragments [Cin/8][16][cout_pad][8] of kfn_conv2d_winograd_s2
    (3x3 stride-2 conv as four stride-1 polyphase filters under F(2,2), csrc/kfn_wino_s2.hip): with
    G = [[1,0],[1,1],[0,1]] fragments 0-8 = (G g00 G^T)[xi][nu] of the 2x2 taps g00[a][b] = w[2a][2b], 9-11 =
    G (w[0][1], w[2][1]), 12-14 = G (w[1][0], w[1][2]), 15 = w[1][1]; every Winograd index 2 carries a minus sign
    (its output coefficient A^T[1][2] = -1 is folded into the weights so that positions can share accumulators)."""
    w = np.asarray(w, np.float32)
    kh, kw, ci, co = w.shape
    assert kh == 3 and kw == 3 and ci % 8 == 0
    G = np.array([[1, 0], [1, 1], [0, 1]], np.float32)
    sg = np.array([1, 1, -1], np.float32)
    u = np.zeros((16, ci, co), np.float32)
    g00 = w[0::2, 0::2]                                        # [2,2,ci,co]
    u00 = np.einsum('xa,abio,nb->xnio', G, g00, G)             # [3,3,ci,co]
    u[0:9] = (u00 * sg[:, None, None, None] * sg[None, :, None, None]).reshape(9, ci, co)
    u[9:12] = np.einsum('xa,aio->xio', G, w[0::2, 1]) * sg[:, None, None]
    u[12:15] = np.einsum('nb,bio->nio', G, w[1, 0::2]) * sg[:, None, None]
    u[15] = w[1, 1]
    cp = -(-co // 32) * 32
    out = np.zeros((16, cp, ci), np.float32)
    out[:, :co, :] = u.transpose(0, 2, 1)
    return np.ascontiguousarray(out.reshape(16, cp, ci // 8, 8).transpose(2, 0, 1, 3))


def pack_winograd_s2_kernel_b(w):
    """The same 16 fragments for the EIGHT-WAVE form of kfn_conv2d_winograd_s2 (kfn_conv_desc.wino_form =
    KFN_WINO_FORM_S2_EIGHT_WAVE, wino_s2b_kernel): [Cin/8][8 fragment pairs][cout_pad][4 k][2 fragments][2 k-steps] -- lane
    (channel n, k) of a 16x16x4 B operand reads 16 contiguous bytes per (chunk, pair): fragments 2q and 2q + 1, input channels
    2k and 2k + 1."""
    u = pack_winograd_s2_kernel(w)                      # [Cin/8][16][cp][8]
    nc, _, cp, _ = u.shape
    v = u.reshape(nc, 8, 2, cp, 4, 2)                   # [chunk][pair][f][n][k][s]
    return np.ascontiguousarray(v.transpose(0, 1, 3, 4, 2, 5).reshape(nc, 8, cp, 16))


def pack_winograd_s2_kernel_c(w):
    """TF HWIO [3,3,Cin,Cout] -> the 36 weight fragments [Cin/16][36][cout_pad][16] of the F(4,2) form of kfn_conv2d_winograd_s2
    (kfn_conv_desc.wino_form = KFN_WINO_FORM_S2_F42, csrc/kfn_wino_s2c.hip: the four stride-1 polyphase filters on 4x4 output
    tiles).  With G = [[1/2,0],[1/2,1/2],[1/6,-1/6],[1/6,1/3],[0,1]] (F(4,2), points {0,1,-1,2,inf}): fragments 0-24 =
    (G g00 G^T)[xi][nu] of the 2x2 taps g00[a][b] = w[2a][2b]; 25-29 = G (w[0][1], w[2][1]) (the (even,odd) phase, transformed
    along y); 30-34 = G (w[1][0], w[1][2]) ((odd,even), along x); 35 = w[1][1].  Transformed in fp64, rounded once to fp32.  Lane
    (channel n, k) of a 16x16x4 B operand reads 16 contiguous bytes per (super-step, fragment): input channels 4k .. 4k+3."""
    w = np.asarray(w, np.float64)
    kh, kw, ci, co = w.shape
    assert kh == 3 and kw == 3 and ci % 16 == 0
    G = np.array([[0.5, 0.0], [0.5, 0.5], [1.0 / 6, -1.0 / 6], [1.0 / 6, 1.0 / 3], [0.0, 1.0]], np.float64)
    u = np.zeros((36, ci, co), np.float64)
    u[0:25] = np.einsum('xa,abio,nb->xnio', G, w[0::2, 0::2], G).reshape(25, ci, co)
    u[25:30] = np.einsum('xa,aio->xio', G, w[0::2, 1])
    u[30:35] = np.einsum('nb,bio->nio', G, w[1, 0::2])
    u[35] = w[1, 1]
    cp = -(-co // 32) * 32
    out = np.zeros((36, cp, ci), np.float32)
    out[:, :co, :] = u.transpose(0, 2, 1).astype(np.float32)
    return np.ascontiguousarray(out.reshape(36, cp, ci // 16, 16).transpose(2, 0, 1, 3))


def as_f16(pack):
    """Wrap a weight packer so that the packed matrix is stored as IEEE halfs (fp16-operand convs)."""
    def f(w):
        return pack(w).astype(np.float16)
    f.__name__ = pack.__name__ + '_f16'
    return f


def as_f16x3(pack):
    """Packer for KFN_OPERAND_F16X3: [hi | lo] half matrices of 1024*w (lo = f16(1024 w - hi))."""
    def f(w):
        m = pack(w).astype(np.float32) * np.float32(1024.0)
        hi = m.astype(np.float16)
        lo = (m - hi.astype(np.float32)).astype(np.float16)
        return np.stack([hi, lo])
    f.__name__ = pack.__name__ + '_f16x3'
    return f


def pack_deconv_kernel(w):
    """TF conv2d_transpose [kh,kw,Cout,Cin] -> [cout_pad][kh*kw*Cin]."""
    kh, kw, co, ci = w.shape
    cp = -(-co // 32) * 32
    out = np.zeros((cp, kh * kw * ci), dtype=np.float32)
    out[:co] = np.transpose(w, (2, 0, 1, 3)).reshape(co, -1)
    return out


def pack_dense_kernel(w):
    """tf.layers.dense [in,out] == a 1x1 conv kernel [1,1,in,out]."""
    return pack_conv_kernel(w.reshape(1, 1, w.shape[0], w.shape[1]))


def pack_first_kernel(w):
    """TF HWIO [3,3,3,C] -> [27][C] (kh,kw,ci major)."""
    return np.ascontiguousarray(w.reshape(27, w.shape[3]).astype(np.float32))


def pack_bias(b):
    return np.ascontiguousarray(b.astype(np.float32))


def split_k_launch_us(workgroups, n_super, k_split, out_bytes, ss_us, fixed_us=10.0, reduce_us=4.0, reduce_bps=4.0e12, cus=256):
    """Cost model (microseconds) of a one-workgroup-per-CU Winograd launch whose input channels are cut into k_split runs:
    rounds of `cus` workgroups x (super-steps of a run x ss_us + the workgroup's prologue / epilogue) + the reduce launch and
    its workspace traffic.  Constants measured at batch 1 and 32 (profiles/r05_*)."""
    rounds = -(-(workgroups * k_split) // cus)
    t = rounds * (-(-n_super // k_split) * ss_us + fixed_us)
    if k_split > 1:
        t += reduce_us + (k_split + 1) * out_bytes / reduce_bps * 1e6
    return t


def best_split_k(workgroups, n_super, out_bytes, ss_us, max_split=8, cus=256, only_below_cus=True):
    """The split that minimises split_k_launch_us; 1 unless splitting wins by more than 10 % (a split launch is two launches
    and a workspace round trip).  only_below_cus: never split a launch that already has a workgroup for every CU."""
    if n_super < 2 or max_split < 2 or (only_below_cus and workgroups >= cus):
        return 1
    base = split_k_launch_us(workgroups, n_super, 1, out_bytes, ss_us, cus=cus)
    best, best_t = 1, base
    for ks in range(2, min(max_split, n_super) + 1):
        if (ks - 1) * (-(-n_super // ks)) >= n_super:     # an empty last run
            continue
        t = split_k_launch_us(workgroups, n_super, ks, out_bytes, ss_us, cus=cus)
        if t < best_t:
            best, best_t = ks, t
    return best if best_t < 0.9 * base else 1


class ConvOp(Op):
    def __init__(self, name, x, y, kernel, bias, kh, kw, stride, relu, transposed=False,
                 epilogue=_lib.EPI_NONE, config=_lib.CFG_AUTO, operand_dtype=_lib.OPERAND_F32):
        self.name = name
        self.x, self.y, self.kernel, self.bias = x, y, kernel, bias
        self.kh, self.kw, self.stride, self.relu = kh, kw, stride, relu
        self.transposed = transposed
        self.epilogue = epilogue
        self.config = config
        self.operand_dtype = operand_dtype
        self.k_step = 0           # 0 = the library's choice
        self._desc = None

    def desc(self):
        n, h, w, cin = self.x.shape
        cout = self.y.shape[3]
        d = _lib.ConvDesc(N=_scaled(n, self.x.graph), H=h, W=w, Cin=cin, ldx=self.x.ld, Cout=cout,
                          cout_pad=-(-cout // 32) * 32, ldy=self.y.ld, kh=self.kh, kw=self.kw,
                          stride=self.stride, transposed=int(self.transposed), relu=int(self.relu),
                          epilogue=self.epilogue, config=self.config, operand_dtype=self.operand_dtype,
                          x_dtype=_lib.ACT_F16 if self.x.dtype == 'f16' else _lib.ACT_F32,
                          y_dtype=_lib.ACT_F16 if self.y.dtype == 'f16' else _lib.ACT_F32, k_step=self.k_step,
                          weights_path=int(getattr(self.x.graph, 'conv_weights_path', 0)),
                          x_layout=_lib.LAYOUT_C16 if self.x.layout == 'c16' else _lib.LAYOUT_NHWC,
                          y_layout=_lib.LAYOUT_C16 if self.y.layout == 'c16' else _lib.LAYOUT_NHWC)
        return d

    def takes_c16(self):
        """Can this op's launch read AND write KFN_LAYOUT_C16 activations?  (Graph.assign_layouts; only two kernels can.)"""
        return False

    def flops(self):
        """Nominal dense FLOPs (zero-padding taps counted)."""
        if self.transposed:
            n, hi, wi, _ = self.x.shape
            return 2.0 * n * hi * wi * self.y.shape[3] * self.kh * self.kw * self.x.shape[3]
        n, ho, wo, cout = self.y.shape
        return 2.0 * n * ho * wo * cout * self.kh * self.kw * self.x.shape[3]

    CFG_TILE = {1: (5, 1, 1, 4), 2: (2, 2, 2, 2), 3: (2, 1, 2, 2), 4: (1, 1, 4, 1), 5: (1, 1, 2, 2),
                6: (2, 1, 4, 1), 7: (3, 1, 2, 2), 8: (5, 1, 1, 8), 9: (2, 4, 2, 2),
                10: (2, 1, 4, 1), 11: (1, 1, 4, 1), 12: (2, 2, 4, 1),   # 10/11: 16-column variants (PREC tag 3)
                13: (4, 4, 2, 2), 14: (4, 2, 2, 4), 15: (4, 2, 4, 1)}   # 256x256: four / eight waves, 512x64 (fp16 activations)
    # template tag PREC of conv_mfma_kernel for fp16 activations: (x is f16, y is f16) -> 4 / 5 / 6
    PREC_F16_IO = {(True, False): 4, (False, True): 5, (True, True): 6}

    def kernel_name(self, lib):
        """Template instantiation this op launches, spelled like rocprofv3 prints it."""
        d = self.desc()
        cfg, bk, tiles = C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.kfn_conv2d_plan(C.byref(d), C.byref(cfg), C.byref(bk), C.byref(tiles)), 'kfn_conv2d_plan')
        t = self.CFG_TILE[cfg.value]
        io = (self.x.dtype == 'f16', self.y.dtype == 'f16')
        if self.operand_dtype == _lib.OPERAND_F16 and io != (False, False):
            prec = self.PREC_F16_IO[io]
            if io == (True, True) and bk.value == 16 and cfg.value in (_lib.CFG_128x256, _lib.CFG_256x256):
                prec = 7      # weights global -> LDS directly (mirror of the AUTO rule in kfn_conv2d_nhwc)
            if io == (True, True) and bk.value == 16 and cfg.value == _lib.CFG_256x256_W8:
                prec = 8      # ... and the activation tile too (the eight-wave tile's AUTO rule)
            return 'conv_mfma_kernel<%d, %d, %d, %d, %d, 0, %d>' % (t + (bk.value if io[1] else 16, prec))
        if self.operand_dtype == _lib.OPERAND_F16:
            return 'conv_mfma_kernel<%d, %d, %d, %d, 16, %d, 1>' % (t + (1 if self.transposed else 0,))
        if self.operand_dtype == _lib.OPERAND_F16X3:
            return 'conv_mfma_kernel<%d, %d, %d, %d, 16, 0, 2>' % t
        return 'conv_mfma_kernel<%d, %d, %d, %d, %d, %d, %d>' % (t + (bk.value, 1 if self.transposed else 0,
                                                                      3 if cfg.value in (10, 11) else 0))

    def workgroups(self, lib=None):
        """Workgroups of this op's launch (tiles of the implicit GEMM: kfn_conv2d_plan)."""
        d = self.desc()
        cfg, bk, tiles = C.c_int(), C.c_int(), C.c_int()
        _lib.check((lib or _lib.load()).kfn_conv2d_plan(C.byref(d), C.byref(cfg), C.byref(bk), C.byref(tiles)), 'kfn_conv2d_plan')
        return tiles.value

    def mfma_flops(self):
        """FLOPs the MFMAs execute; for the direct kernel the nominal count (tile padding not counted)."""
        return self.flops()

    def launch(self, lib, stream):
        d = self.desc()
        rc = lib.kfn_conv2d_nhwc(C.byref(d), self.x.ptr, self.kernel.ptr,
                                 self.bias.ptr if self.bias is not None else None, self.y.ptr, stream)
        _lib.check(rc, 'kfn_conv2d_nhwc[%s]' % self.name)


class WindowFcConvOp(ConvOp):
    """3x3 stride-1 SAME conv on 2x2 images (OFlowNet's bottleneck level) as one dense [4 Cin] x [4 Cout] matrix per
    window through the 1x1 path of kfn_conv2d_nhwc (pack_window_fc_kernel): 16/36 of the nine-tap form's MFMAs.
    Needs both tensors pixel-contiguous; `resolve()` (called by Graph.finalize, after every concat has re-bound its
    producers) falls back to the plain convolution otherwise."""

    def __init__(self, name, x, y, kernel, bias, relu):
        ConvOp.__init__(self, name, x, y, kernel, bias, 3, 3, 1, relu)

    def resolve(self):
        if self.x.ld == self.x.shape[3] and self.y.ld == self.y.shape[3]:
            return
        if self.kernel.storage is not None:
            raise _lib.KfnError('%s: weights already packed as a window matrix' % self.name)
        self.kernel.pack = as_f16(pack_conv_kernel) if self.operand_dtype == _lib.OPERAND_F16 else pack_conv_kernel
        if self.bias is not None:
            self.bias.pack = pack_bias
        self.__class__ = ConvOp

    def desc(self):
        n, h, w, cin = self.x.shape
        cout = self.y.shape[3]
        assert h == 2 and w == 2 and self.x.ld == cin and self.y.ld == cout
        return _lib.ConvDesc(N=_scaled(n, self.x.graph), H=1, W=1, Cin=4 * cin, ldx=4 * cin, Cout=4 * cout,
                             cout_pad=-(-(4 * cout) // 32) * 32, ldy=4 * cout, kh=1, kw=1, stride=1, transposed=0,
                             relu=int(self.relu), epilogue=self.epilogue, config=self.config,
                             operand_dtype=self.operand_dtype)

    def mfma_flops(self):
        n, h, w, cin = self.x.shape
        return 2.0 * _scaled(n, self.x.graph) * (4 * cin) * (4 * self.y.shape[3])


class WinogradConvOp(ConvOp):
    """3x3 stride-1 SAME conv through kfn_conv2d_winograd (2.25x fewer MFMA FLOPs)."""

    def __init__(self, name, x, y, kernel, bias, relu, workspace):
        ConvOp.__init__(self, name, x, y, kernel, bias, 3, 3, 1, relu)
        self.workspace = workspace  # Storage shared by all Winograd layers of the graph

    def kernel_name(self, lib):
        d = self.desc()
        cfg, bk, tiles = C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.kfn_winograd_plan(C.byref(d), C.byref(cfg), C.byref(bk), C.byref(tiles)), 'kfn_winograd_plan')
        return 'conv_mfma_kernel<%d, %d, %d, %d, %d, 2, 0>' % (self.CFG_TILE[cfg.value] + (bk.value,))

    def mfma_flops(self):
        """FLOPs the 16 GEMMs actually execute (algorithmic flops() stays the nominal 2*M*N*K)."""
        n, ho, wo, cout = self.y.shape
        return 2.0 * 16 * n * ((ho + 1) // 2) * ((wo + 1) // 2) * cout * self.x.shape[3]

    def workspace_bytes(self):
        n, ho, wo, cout = self.y.shape
        return 16 * n * ((ho + 1) // 2) * ((wo + 1) // 2) * cout * 4

    def launch(self, lib, stream, phases=3):
        d = self.desc()
        rc = lib.kfn_conv2d_winograd(C.byref(d), self.x.ptr, self.kernel.ptr,
                                     self.bias.ptr if self.bias is not None else None, self.y.ptr,
                                     self.workspace.ptr, phases, stream)
        _lib.check(rc, 'kfn_conv2d_winograd[%s]' % self.name)


class WinogradFusedConvOp(ConvOp):
    """3x3 stride-1 SAME conv through kfn_conv2d_winograd_fused: all 16 Winograd positions of a tile
    block in registers, input and output transforms in the kernel -- one launch, no workspace.  The library
    picks the form: four waves sharing one input transform through LDS (wino3_kernel, 128 output channels
    per workgroup) when Cout >= 128 and Cin % 32 == 0, two waves sharing it (wino3_pair_kernel) for 33 .. 64 output
    channels with Cin % 16 == 0, else one wave per 32 output channels (wino2_kernel)."""

    def __init__(self, name, x, y, kernel, bias, relu, operand_dtype=_lib.OPERAND_F32):
        ConvOp.__init__(self, name, x, y, kernel, bias, 3, 3, 1, relu, operand_dtype=operand_dtype)

    @staticmethod
    def supported(x_shape, cin, cout, operand_dtype=_lib.OPERAND_F32):
        n, h, w, _ = x_shape
        if operand_dtype == _lib.OPERAND_F16:      # fp16 operands: the four-wave form only
            return cin % 64 == 0 and cout >= 128 and (h + 1) // 2 >= 4
        return cin % 16 == 0 and (h + 1) // 2 >= 4

    def four_wave(self):
        """Mirror of the routing rule in kfn_conv2d_winograd_fused (kfn_wino2.hip)."""
        return self.y.shape[3] >= 128 and self.x.shape[3] % 32 == 0

    def two_wave(self):
        """The second routing rule of kfn_conv2d_winograd_fused: all of a layer's 33 .. 64 output channels in one
        workgroup of two waves (two images of the input below 1 GiB: the patch offsets carry their marks above)."""
        n, h, w, cin = self.x.shape
        return (not self.four_wave() and self.operand_dtype == _lib.OPERAND_F32 and 32 < self.y.shape[3] <= 64
                and cin % 16 == 0 and 2 * h * w * self.x.ld * 4 < (1 << 30))

    def kernel_name(self, lib):
        if self.operand_dtype == _lib.OPERAND_F16:
            return 'wino3_kernel<true>'
        if self.two_wave():
            return 'wino3_pair_kernel'
        return 'wino3_kernel' if self.four_wave() else 'wino2_kernel'

    def mfma_flops(self):
        """FLOPs the MFMAs execute: 16 positions x (tile blocks padded to 8x4 tiles, batch rows packed) x
        output channels padded to the workgroup's column block."""
        n, ho, wo, cout = self.y.shape
        n = _scaled(n, self.x.graph)
        th, tw = (ho + 1) // 2, (wo + 1) // 2
        tiles = (-(-tw // 8) * 8) * (-(-(n * th) // 4) * 4)
        cpad = -(-cout // 32) * 32
        if self.four_wave():
            cpad = -(-cpad // 128) * 128
        return 2.0 * 16 * tiles * cpad * self.x.shape[3]

    def workgroups(self, lib=None):
        """Blocks of 8 x 4 tiles of 2x2 pixels (the tile rows of the batch packed) x column blocks of 128 (four waves), 64
        (two waves: the whole layer) or 32 (one wave) output channels."""
        n, ho, wo, cout = self.y.shape
        n = _scaled(n, self.x.graph)
        th, tw = (ho + 1) // 2, (wo + 1) // 2
        blocks = (-(-tw // 8)) * (-(-(n * th) // 4))
        cpad = -(-cout // 32) * 32
        per = 128 if (self.four_wave() or self.operand_dtype == _lib.OPERAND_F16) else (64 if self.two_wave() else 32)
        return blocks * (-(-cpad // per))

    def launch(self, lib, stream, phases=3):
        d = self.desc()
        rc = lib.kfn_conv2d_winograd_fused(C.byref(d), self.x.ptr, self.kernel.ptr,
                                           self.bias.ptr if self.bias is not None else None, self.y.ptr, stream)
        _lib.check(rc, 'kfn_conv2d_winograd_fused[%s]' % self.name)


class WinogradF43ConvOp(ConvOp):
    """3x3 stride-1 SAME conv through kfn_conv2d_winograd_f43 (csrc/kfn_wino4.hip): F(4x4,3x3), one launch, no workspace.
    eight_wave (Graph.winograd_f43_eight_wave, the default): wino4b_kernel -- two waves per SIMD on 16x16x4 MFMA tiles, 36
    accumulators of 4 registers per wave, weights packed per pair of positions (pack_winograd_f43_kernel_b); else
    wino4_kernel -- four waves on 32x32x2 tiles, 18 accumulators of 16 registers (pack_winograd_f43_kernel)."""

    def __init__(self, name, x, y, kernel, bias, relu, eight_wave=True, k_split=1, workspace=None):
        ConvOp.__init__(self, name, x, y, kernel, bias, 3, 3, 1, relu)
        self.eight_wave = eight_wave
        self.k_split = int(k_split)      # > 1: kfn_conv2d_winograd_f43_splitk (eight-wave form only) with the graph's shared workspace
        self.workspace = workspace

    SS_US = 3.8      # one super-step (16 input channels) of a workgroup: 144 MFMAs of 32 cycles on each of two waves per SIMD

    @classmethod
    def launch_us(cls, workgroups, n_super, k_split, out_bytes, cus=256):
        return split_k_launch_us(workgroups, n_super, k_split, out_bytes, cls.SS_US, cus=cus)

    @classmethod
    def best_k_split(cls, workgroups, cin, out_bytes, max_split=8, cus=256):
        """Split of the input channels for a launch of fewer workgroups than CUs (single frames: conv4b 160, conv5 80, conv6
        40 workgroups on 256 CUs)."""
        return best_split_k(workgroups, cin // 16, out_bytes, cls.SS_US, max_split, cus)

    def desc(self):
        d = ConvOp.desc(self)
        d.wino_form = _lib.WINO_FORM_F43_EIGHT_WAVE if self.eight_wave else _lib.WINO_FORM_F43_FOUR_WAVE
        return d

    def takes_c16(self):
        return type(self) is WinogradF43ConvOp and self.eight_wave and self.k_split <= 1

    @staticmethod
    def supported(x_shape, cin, cout, ldx=None, ldy=None, y_ch_off=0):
        """Mirror of kfn_winograd_f43_supported (+ the launcher's 16-byte alignment of y, which for a tensor is "pixel
        stride and channel offset multiples of 4": a concat slice at an odd offset must take another route -- the fused
        F(2x2) and stride-2 kernels have a dword-store path for it, this kernel has not)."""
        n, h, w, _ = x_shape
        ldx = cin if ldx is None else ldx
        ldy = cout if ldy is None else ldy
        return (cin % 16 == 0 and (h + 3) // 4 >= 8 and cout % 4 == 0 and ldx % 2 == 0 and ldx >= cin and ldy >= cout
                and ldy % 4 == 0 and y_ch_off % 4 == 0
                and 2 * h * w * ldx * 4 < (1 << 30) and 2 * h * w * ldy * 4 < (1 << 31)
                and 36 * (-(-cout // 32) * 32) * cin * 4 < (1 << 31))

    def resolve(self):
        """Graph.finalize: a later concat may have re-bound the output into a wider buffer; if the F(4x4,3x3) launcher
        would now reject it, fall back to the fused F(2x2,3x3) kernel (dword stores for unaligned outputs) or the direct
        one -- the weights are not packed yet."""
        n, h, w, cin = self.x.shape
        if self.supported(self.x.shape, cin, self.y.shape[3], self.x.ld, self.y.ld, self.y.ch_off):
            return
        if self.kernel.storage is not None:
            raise _lib.KfnError('%s: weights already packed for the F(4x4,3x3) kernel' % self.name)
        self.__dict__.pop('eight_wave', None)
        self.__dict__.pop('k_split', None)
        self.workspace = None
        if WinogradFusedConvOp.supported(self.x.shape, cin, self.y.shape[3]) and min(h, w) >= 8:
            self.kernel.pack = pack_winograd_fused_kernel
            self.__class__ = WinogradFusedConvOp
        else:
            self.kernel.pack = pack_conv_kernel
            self.__class__ = ConvOp

    @staticmethod
    def workgroups(x_shape, cout):
        """Workgroups of the launch: blocks of 4 x 8 tiles (the tile rows of the batch's images packed) x 64 channels."""
        n, h, w, _ = x_shape
        th, tw = (h + 3) // 4, (w + 3) // 4
        return (-(-tw // 4)) * (-(-(n * th) // 8)) * (-(-cout // 64))

    def launch_workgroups(self):
        n, h, w, _ = self.x.shape
        return self.workgroups((_scaled(n, self.x.graph), h, w, 0), self.y.shape[3]) * max(1, self.k_split)

    def kernel_name(self, lib):
        if self.k_split > 1:
            return 'wino4b_kernel[split-K %d] + wino4_splitk_reduce_kernel' % self.k_split
        return 'wino4b_kernel' if self.eight_wave else 'wino4_kernel'

    def mfma_flops(self):
        """FLOPs the MFMAs execute: 36 positions x (tile blocks padded to 4x8 tiles of 4x4 pixels, batch rows packed) x
        output channels padded to the workgroup's 64."""
        n, ho, wo, cout = self.y.shape
        n = _scaled(n, self.x.graph)
        th, tw = (ho + 3) // 4, (wo + 3) // 4
        tiles = (-(-tw // 4) * 4) * (-(-(n * th) // 8) * 8)
        cpad = -(-cout // 64) * 64
        return 2.0 * 36 * tiles * cpad * self.x.shape[3]

    def workspace_bytes(self):
        n, ho, wo, cout = self.y.shape
        return self.k_split * _scaled(n, self.x.graph) * ho * wo * cout * 4 if self.k_split > 1 else 0

    def launch(self, lib, stream, phases=3):
        d = self.desc()
        bias = self.bias.ptr if self.bias is not None else None
        if self.k_split > 1:
            rc = lib.kfn_conv2d_winograd_f43_splitk(C.byref(d), self.x.ptr, self.kernel.ptr, bias, self.y.ptr,
                                                    self.workspace.ptr, self.k_split, stream)
            _lib.check(rc, 'kfn_conv2d_winograd_f43_splitk[%s]' % self.name)
            return
        rc = lib.kfn_conv2d_winograd_f43(C.byref(d), self.x.ptr, self.kernel.ptr, bias, self.y.ptr, stream)
        _lib.check(rc, 'kfn_conv2d_winograd_f43[%s]' % self.name)


def pack_conv64_rows_kernel(w):
    """TF HWIO [3,3,64,64] -> the A fragments of kfn_conv3x3_c64_f16 (csrc/kfn_conv64.hip): [2 channel halves][36][64 lanes][8]
    halfs -- fragment f = (dy*3 + dx)*4 + c of lane l (row i = l % 32, k half hk = l // 32) holds
    w[dy][dx][16c + 8hk + t][32 half + i], t = 0..7 (v_mfma_f32_32x32x16_f16 with the WEIGHTS as the A operand)."""
    w = np.asarray(w, np.float32)
    assert w.shape == (3, 3, 64, 64)
    wt = w.reshape(3, 3, 4, 2, 8, 2, 32)               # [dy][dx][c][hk][t][half][i]
    out = wt.transpose(5, 0, 1, 2, 3, 6, 4)            # [half][dy][dx][c][hk][i][t]
    return np.ascontiguousarray(out.reshape(2, 36, 64, 8)).astype(np.float16)


class Conv64RowsF16Op(ConvOp):
    """3x3 stride-1 SAME conv 64 -> 64 on fp16 activations through kfn_conv3x3_c64_f16 (csrc/kfn_conv64.hip): weights
    resident in registers, every input row read from LDS once for the three output rows it feeds -- SCoordNet's conv1b in
    BASELINE config 5."""

    def __init__(self, name, x, y, kernel, bias, relu):
        ConvOp.__init__(self, name, x, y, kernel, bias, 3, 3, 1, relu, operand_dtype=_lib.OPERAND_F16)

    @staticmethod
    def supported(x, y, cin, cout):
        n, h, w, _ = x.shape
        return (cin == 64 and cout == 64 and x.dtype == 'f16' and y.dtype == 'f16' and x.ld % 8 == 0 and y.ld % 8 == 0
                and h * w * max(x.ld, y.ld) * 2 < (1 << 31))

    def kernel_name(self, lib):
        w = self.x.shape[2]
        return 'conv64_rows_kernel<%d, %s>' % (3 if -(-w // 192) * 192 <= -(-w // 128) * 128 else 2, 'true' if self.relu else 'false')

    def mfma_flops(self):
        """FLOPs the MFMAs execute: the strips' padded width, plus two halo-row steps per chunk of rows (not counted: the
        chunk count is the launcher's choice) -- the nominal FLOPs scaled by the width padding."""
        w = self.x.shape[2]
        wp = min(-(-w // 192) * 192, -(-w // 128) * 128)
        return self.flops() * wp / float(w)

    def launch(self, lib, stream, phases=3):
        d = self.desc()
        rc = lib.kfn_conv3x3_c64_f16(C.byref(d), self.x.ptr, self.kernel.ptr,
                                     self.bias.ptr if self.bias is not None else None, self.y.ptr, stream)
        _lib.check(rc, 'kfn_conv3x3_c64_f16[%s]' % self.name)


class WinogradS2ConvOp(ConvOp):
    """3x3 stride-2 SAME conv of an even-sized image through kfn_conv2d_winograd_s2 (polyphase + F(2,2):
    25 MFMA streams into 9 accumulators per 2x2 outputs instead of 36 direct taps)."""

    SS_US = 5.3      # one super-step of wino_s2b_kernel: 200 MFMAs of 32 cycles on each of two waves per SIMD

    def __init__(self, name, x, y, kernel, bias, relu, operand_dtype=_lib.OPERAND_F32, eight_wave=False, k_split=1, workspace=None,
                 f42=False):
        ConvOp.__init__(self, name, x, y, kernel, bias, 3, 3, 2, relu, operand_dtype=operand_dtype)
        # wino_s2c_kernel: polyphase + F(4,2) on 4x4 output tiles, 81 instead of 100 products per 16 outputs (fp32; H, W multiples
        # of 8; weights pack_winograd_s2_kernel_c).  Blocks of 16x16 output pixels: for launches that fill the chip several times.
        self.f42 = bool(f42) and operand_dtype == _lib.OPERAND_F32
        if self.f42:
            eight_wave, k_split = False, 1
        self.k_split = int(k_split)      # > 1: kfn_conv2d_winograd_s2_splitk (eight-wave form, fp32) with a private workspace
        self.workspace = workspace
        # wino_s2b_kernel (two waves per SIMD on 16x16x4 MFMA tiles; fp32 operands; weights packed per pair of fragments,
        # pack_winograd_s2_kernel_b) instead of wino_s2_kernel
        self.eight_wave = bool(eight_wave) and operand_dtype == _lib.OPERAND_F32

    def desc(self):
        d = ConvOp.desc(self)
        if self.f42:
            d.wino_form = _lib.WINO_FORM_S2_F42
        elif self.eight_wave:
            d.wino_form = _lib.WINO_FORM_S2_EIGHT_WAVE
        return d

    def takes_c16(self):
        return type(self) is WinogradS2ConvOp and self.f42

    @staticmethod
    def supported(x_shape, cin, cout):
        n, h, w, _ = x_shape
        return cin % 16 == 0 and h % 2 == 0 and w % 2 == 0 and (h // 2 + 1) // 2 >= 4

    @staticmethod
    def f42_supported(x_shape, cin, cout, ldy=None, y_ch_off=0):
        """kfn::wino_s2c_supported's pointer-free conditions (csrc/kfn_wino_s2c.hip)."""
        n, h, w, _ = x_shape
        ldy = cout if ldy is None else ldy
        return (cin % 16 == 0 and h % 8 == 0 and w % 8 == 0 and h >= 32 and cout % 4 == 0 and ldy % 4 == 0 and y_ch_off % 4 == 0)

    @staticmethod
    def f42_workgroups(y_shape):
        """Blocks of 4 x 4 tiles of 4x4 OUTPUT pixels (batch rows packed) x column blocks of 128 output channels."""
        n, ho, wo, cout = y_shape
        th, tw = -(-ho // 4), -(-wo // 4)
        return (-(-tw // 4)) * (-(-(n * th) // 4)) * (-(-cout // 128))

    def resolve(self):
        """After every concat has re-bound its producers: the F(4,2) form and the split-K form store 16 bytes at a time."""
        y = self.y
        if self.f42 and not self.f42_supported(self.x.shape, self.x.shape[3], y.shape[3], y.ld, y.ch_off):
            if self.kernel.storage is not None:
                raise _lib.KfnError('%s: output window (ld %d, channel offset %d) cannot take the F(4,2) stride-2 form and its weights '
                                    'are already uploaded in that layout' % (self.name, y.ld, y.ch_off))
            self.f42, self.eight_wave = False, True          # the F(2,2) eight-wave form has a dword-store path
            self.kernel.pack = pack_winograd_s2_kernel_b
        if self.k_split > 1 and (y.ld % 4 != 0 or y.ch_off % 4 != 0):
            #  kfn_conv2d_winograd_s2_splitk needs a 16-byte aligned output with ldy % 4 == 0; the plain eight-wave
            # launch takes the dword-store path instead
            self.k_split = 1
            if self.workspace is not None and self.workspace in self.x.graph.storages:
                self.x.graph.storages.remove(self.workspace)
            self.workspace = None

    def kernel_name(self, lib):
        if self.f42:
            # (the launcher's rule, csrc/kfn_wino_s2c.hip: the persistent form for 2 .. 64 super-steps and at least two workgroups per CU)
            ss = self.x.shape[3] // 16
            return 'wino_s2c_pkernel' if 2 <= ss <= 64 and self.workgroups() >= 2 * getattr(self.x.graph, 'cu_count', 256) else 'wino_s2c_kernel'
        if self.k_split > 1:
            return 'wino_s2b_kernel[split-K %d] + splitk_reduce_kernel' % self.k_split
        if self.eight_wave:
            return 'wino_s2b_kernel'
        return 'wino_s2_kernel<true>' if self.operand_dtype == _lib.OPERAND_F16 else 'wino_s2_kernel'

    def mfma_flops(self):
        """FLOPs the MFMAs execute: 25 products per 2x2-output tile and input channel, tile blocks padded to
        8x4 tiles (batch rows packed), output channels padded to the workgroup's 128."""
        n, ho, wo, cout = self.y.shape
        n = _scaled(n, self.x.graph)
        if self.f42:      # 81 products per 4x4-output tile and input channel, blocks of 4x4 tiles
            th, tw = -(-ho // 4), -(-wo // 4)
            tiles = (-(-tw // 4) * 4) * (-(-(n * th) // 4) * 4)
            return 2.0 * 81 * tiles * (-(-cout // 128) * 128) * self.x.shape[3]
        th, tw = (ho + 1) // 2, (wo + 1) // 2
        tiles = (-(-tw // 8) * 8) * (-(-(n * th) // 4) * 4)
        return 2.0 * 25 * tiles * (-(-cout // 128) * 128) * self.x.shape[3]

    def workgroups(self, lib=None):
        """Blocks of 8 x 4 tiles of 2x2 OUTPUT pixels (batch rows packed) x column blocks of 128 output channels."""
        n, ho, wo, cout = self.y.shape
        n = _scaled(n, self.x.graph)
        if self.f42:
            return self.f42_workgroups((n, ho, wo, cout))
        th, tw = (ho + 1) // 2, (wo + 1) // 2
        return (-(-tw // 8)) * (-(-(n * th) // 4)) * (-(-cout // 128)) * max(1, self.k_split)

    @staticmethod
    def base_workgroups(y_shape):
        n, ho, wo, cout = y_shape
        th, tw = (ho + 1) // 2, (wo + 1) // 2
        return (-(-tw // 8)) * (-(-(n * th) // 4)) * (-(-cout // 128))

    @classmethod
    def best_k_split(cls, workgroups, cin, out_bytes, max_split=8, cus=256):
        """Unlike the F(4x4) rule this one also splits launches a little ABOVE one round (conv4a at batch 1: 320 workgroups =
        two rounds, the second a quarter full) -- up to two rounds."""
        if workgroups >= 2 * cus:
            return 1
        return best_split_k(workgroups, cin // 16, out_bytes, cls.SS_US, max_split, cus, only_below_cus=False)

    def workspace_bytes(self):
        n, ho, wo, cout = self.y.shape
        return self.k_split * _scaled(n, self.x.graph) * ho * wo * cout * 4 if self.k_split > 1 else 0

    def launch(self, lib, stream, phases=3):
        d = self.desc()
        bias = self.bias.ptr if self.bias is not None else None
        if self.k_split > 1:
            rc = lib.kfn_conv2d_winograd_s2_splitk(C.byref(d), self.x.ptr, self.kernel.ptr, bias, self.y.ptr, self.workspace.ptr,
                                                   self.k_split, stream)
            _lib.check(rc, 'kfn_conv2d_winograd_s2_splitk[%s]' % self.name)
            return
        rc = lib.kfn_conv2d_winograd_s2(C.byref(d), self.x.ptr, self.kernel.ptr, bias, self.y.ptr, stream)
        _lib.check(rc, 'kfn_conv2d_winograd_s2[%s]' % self.name)


class FirstConvOp(Op):
    """uint8 image -> preprocess -> conv (Cin=3), up to two heads sharing one image read."""

    def __init__(self, img):
        self.name = 'first_conv'
        self.img = img
        self.heads = []  # (name, y, kernel, bias)

    def add_head(self, name, y, kernel, bias):
        assert len(self.heads) < 2
        self.heads.append((name, y, kernel, bias))
        self.name = 'first_conv[' + '+'.join(h[0] for h in self.heads) + ']'

    def flops(self):
        n, h, w, _ = self.img.shape
        return sum(2.0 * n * h * w * 27 * hd[1].shape[3] for hd in self.heads)

    def launch(self, lib, stream):
        n, h, w, _ = self.img.shape
        n = _scaled(n, self.img.graph)
        h1 = self.heads[0]
        for hd in self.heads:
            assert hd[1].is_whole(), 'first-layer outputs must be whole buffers'
        # an fp16 head (BASELINE config 5: SCoordNet conv1a) must be the first one
        if len(self.heads) == 2 and self.heads[1][1].dtype == 'f16':
            self.heads.reverse()
            h1 = self.heads[0]
        dt1 = _lib.ACT_F16 if h1[1].dtype == 'f16' else _lib.ACT_F32
        if len(self.heads) == 2:
            h2 = self.heads[1]
            assert h2[1].dtype == 'f32', 'only one first-layer head can write fp16'
            rc = lib.kfn_first_conv_u8_ex(self.img.ptr, n, h, w, h1[2].ptr, h1[3].ptr, h1[1].ptr, h1[1].shape[3], dt1,
                                          h2[2].ptr, h2[3].ptr, h2[1].ptr, h2[1].shape[3], stream)
        else:
            rc = lib.kfn_first_conv_u8_ex(self.img.ptr, n, h, w, h1[2].ptr, h1[3].ptr, h1[1].ptr, h1[1].shape[3], dt1,
                                          None, None, None, 0, stream)
        _lib.check(rc, 'kfn_first_conv_u8')


class CostVolumeOp(Op):
    def __init__(self, f1, f2, vol, window):
        self.name = 'cost_volume'
        self.f1, self.f2, self.vol, self.window = f1, f2, vol, window

    def launch(self, lib, stream):
        n, h, w, c = self.f2.shape
        n = _scaled(n, self.f2.graph)
        assert self.f1.ld == c and self.f2.ld == c
        rc = lib.kfn_cost_volume(self.f1.ptr, self.f2.ptr, self.vol.ptr, n, h, w, c, self.window, stream)
        _lib.check(rc, 'kfn_cost_volume')


class CostVolumeConvOp(Op):
    """BuildCoordVolume + OFlowNet conv0 in one MFMA launch (kfn_cost_volume_conv): the cost
    volume is generated in the kernel's loader and never written."""

    def __init__(self, f1, f2, y, kernel, bias, relu, window):
        assert window == 8
        self.name = 'conv0[cost_volume]'
        self.f1, self.f2, self.y, self.kernel, self.bias, self.relu = f1, f2, y, kernel, bias, relu
        self.config = _lib.CFG_AUTO

    def flops(self):
        n, h, w, c = self.f2.shape
        return 2.0 * n * h * w * 64 * 9 * c * self.y.shape[3]

    def kernel_name(self, lib):
        n, h, w, c = self.f2.shape
        d = _lib.ConvDesc(N=n * h * w, H=8, W=8, Cin=c, ldx=c, Cout=self.y.shape[3],
                          cout_pad=-(-self.y.shape[3] // 32) * 32, ldy=self.y.ld, kh=3, kw=3, stride=1, config=self.config)
        cfg, bk, tiles = C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.kfn_conv2d_plan(C.byref(d), C.byref(cfg), C.byref(bk), C.byref(tiles)), 'kfn_conv2d_plan')
        return 'conv_mfma_kernel<%d, %d, %d, %d, 16, 3, 0>' % ConvOp.CFG_TILE[cfg.value]

    def launch(self, lib, stream):
        n, h, w, c = self.f2.shape
        n = _scaled(n, self.f2.graph)
        assert self.f1.ld == c and self.f2.ld == c
        cout = self.y.shape[3]
        rc = lib.kfn_cost_volume_conv(self.f1.ptr, self.f2.ptr, self.kernel.ptr,
                                      self.bias.ptr if self.bias is not None else None, self.y.ptr, n, h, w, c,
                                      cout, -(-cout // 32) * 32, self.y.ld, int(self.relu), self.config, stream)
        _lib.check(rc, 'kfn_cost_volume_conv')


class DerivedConvOp(ConvOp):
    """A convolution that exists only because of an algebraic rewrite (e.g. the class
    convolutions of the factored cost volume): it executes MFMA work but carries none of the
    reference's nominal FLOPs -- those stay with the op that finishes the rewritten layer."""

    def flops(self):
        return 0.0

    def mfma_flops(self):
        return ConvOp.flops(self)


class PadOp(Op):
    """Zero-pad an NHWC map by `pad` pixels on every side (kfn_pad_nhwc)."""

    def __init__(self, x, y, pad):
        self.name = 'pad[%s]' % (x.name or '?')
        self.x, self.y, self.pad = x, y, pad

    def kernel_name(self, lib):
        return 'pad_nhwc_kernel'

    def launch(self, lib, stream):
        n, h, w, c = self.x.shape
        n = _scaled(n, self.x.graph)
        assert self.x.ld == c and self.y.ld == c
        _lib.check(lib.kfn_pad_nhwc(self.x.ptr, self.y.ptr, n, h, w, c, self.pad, stream), 'kfn_pad_nhwc')


class CostVolumeGatherOp(Op):
    """Last step of the factored cost volume + conv0 (kfn_cost_volume_gather):
    y[p,ci,cj] = act(T_k[p] - G_k[p + (ci-4, cj-4)])."""

    def __init__(self, t, gp, y, relu, cin):
        self.cin = cin   # channels of the feature maps (conv0's Cin)
        self.name = 'conv0[cost_volume gather]'
        self.t, self.gp, self.y, self.relu = t, gp, y, relu

    def kernel_name(self, lib):
        return 'cost_volume_gather_kernel'

    def flops(self):
        """The nominal FLOPs of the layer this op completes: conv0 on every window cell."""
        n, h, w, c9 = self.t.shape
        co = self.y.shape[3]
        return 2.0 * n * h * w * 64 * 9 * self.cin * co

    def mfma_flops(self):
        return 0.0

    def launch(self, lib, stream):
        n, h, w, c9 = self.t.shape
        n = _scaled(n, self.t.graph)
        c = self.y.shape[3]
        assert c9 == 9 * c and self.t.ld == c9 and self.gp.ld == c9
        _lib.check(lib.kfn_cost_volume_gather(self.t.ptr, self.gp.ptr, self.y.ptr, n, h, w, c, self.y.ld,
                                              int(self.relu), stream), 'kfn_cost_volume_gather')


class FlowOp(Op):
    def __init__(self, logits, flow, prob, window):
        self.name = 'flow_softargmax'
        self.logits, self.flow, self.prob, self.window = logits, flow, prob, window

    def launch(self, lib, stream):
        P = _scaled(self.flow.pixels, self.flow.graph)
        rc = lib.kfn_flow_softargmax(self.logits.ptr, self.flow.ptr,
                                     self.prob.ptr if self.prob is not None else None, P, self.window, stream)
        _lib.check(rc, 'kfn_flow_softargmax')


def _cvol_tap_valid(r, k):
    """Does tap k (0..2) of a 3-tap axis stay inside the 8-cell window for a cell of border
    class r (0 = first cell, 1 = interior, 2 = last cell)?  (conv0's own SAME padding)"""
    return not ((r == 0 and k == 0) or (r == 2 and k == 2))


def cvol_class_kernels(w):
    """conv0's TF kernel [3,3,C,Co] -> (W9 [3,3,C,9*Co], S9 [1,1,C,9*Co]) for the factored cost
    volume (include/kfnet_hip.h, kfn_cost_volume_gather): class k = 3*rowclass + colclass keeps
    the taps that stay inside the window; S9 is their sum (the f2 term), W9 the masked kernel
    (the f1 term).  Sums in fp64, like the Winograd weight transform."""
    w = np.asarray(w, dtype=np.float64)
    kh, kw, c, co = w.shape
    assert kh == 3 and kw == 3
    w9 = np.zeros((3, 3, c, 9 * co))
    s9 = np.zeros((1, 1, c, 9 * co))
    for ry in range(3):
        for rx in range(3):
            k = ry * 3 + rx
            for ky in range(3):
                for kx in range(3):
                    if _cvol_tap_valid(ry, ky) and _cvol_tap_valid(rx, kx):
                        w9[ky, kx, :, k * co:(k + 1) * co] = w[ky, kx]
                        s9[0, 0, :, k * co:(k + 1) * co] += w[ky, kx]
    return w9.astype(np.float32), s9.astype(np.float32)


def pack_cvol_g_kernel(w):
    return pack_conv_kernel(cvol_class_kernels(w)[0])


def pack_cvol_t_kernel(w):
    return pack_conv_kernel(cvol_class_kernels(w)[1])


def pack_cvol_bias(b):
    return np.tile(np.asarray(b, dtype=np.float32), 9)


def pack_flow_head_kernel(w):
    """TF HWIO [3,3,C,1] -> [3][3][C]."""
    assert w.shape[0] == 3 and w.shape[1] == 3 and w.shape[3] == 1
    return np.ascontiguousarray(w[..., 0].astype(np.float32))


class FlowHeadOp(Op):
    """OFlowNet 'prediction' conv + softmax + soft-argmax in one launch (kfn_flow_head)."""

    def __init__(self, x, kernel, bias, flow, logits=None):
        self.name = 'flow_head'
        self.x, self.kernel, self.bias, self.flow, self.logits = x, kernel, bias, flow, logits

    def flops(self):
        n, h, w, c = self.x.shape
        return 2.0 * n * h * w * 9 * c

    def launch(self, lib, stream):
        n, h, w, c = self.x.shape
        assert h == 8 and w == 8 and self.x.ld == c
        P = _scaled(n, self.x.graph)
        rc = lib.kfn_flow_head(self.x.ptr, self.kernel.ptr, self.bias.ptr if self.bias is not None else None,
                               self.flow.ptr, self.logits.ptr if self.logits is not None else None, P, c, stream)
        _lib.check(rc, 'kfn_flow_head')


def pack_oflow_tail_kernel(w):
    """TF HWIO [3,3,48,16] (OFlowNet conv6) -> the per-lane weight fragments [108][64] of kfn_oflow_tail:
    fragment t = tap*12 + j of lane (kq = lane // 16, n = lane % 16) is w[tap][kq*12 + j][n] (16x16x4 MFMA B
    operand, K ordered so that a lane's 12 k-steps of a tap are 12 consecutive input channels)."""
    w = np.asarray(w, np.float32)
    assert w.shape == (3, 3, 48, 16)
    wt = w.reshape(9, 4, 12, 16)                       # [tap][kq][j][n]
    return np.ascontiguousarray(wt.transpose(0, 2, 1, 3).reshape(108, 64))


def pack_oflow_tail_kernel_f16(w):
    """conv6 for kfn_oflow_tail2_f16: [27][64][4] halfs -- fragment t = tap*3 + s of lane (kq, n) holds
    w[tap][kq*12 + 4s + j][n], j = 0..3 (the B operand of one v_mfma_f32_16x16x16_f16: the same 12 consecutive input
    channels per lane and tap as pack_oflow_tail_kernel, four to an MFMA)."""
    w = np.asarray(w, np.float32)
    assert w.shape == (3, 3, 48, 16)
    wt = w.reshape(9, 4, 3, 4, 16)                     # [tap][kq][s][j][n]
    return np.ascontiguousarray(wt.transpose(0, 2, 1, 4, 3).reshape(27, 64, 4)).astype(np.float16)


class OFlowTailOp(Op):
    """OFlowNet conv6 + 'prediction' conv + softmax + soft-argmax in one launch (kfn_oflow_tail): one wave per
    window, the 8x8x48 patch resident in LDS, conv6's weights in registers."""

    def __init__(self, x, k6, b6, kp, bpred, flow, logits=None):
        self.name = 'oflow_tail[conv6+prediction+softargmax]'
        self.x, self.k6, self.b6, self.kp, self.bpred, self.flow, self.logits = x, k6, b6, kp, bpred, flow, logits

    def kernel_name(self, lib):
        return 'oflow_tail_kernel'

    def flops(self):
        n, h, w, c = self.x.shape
        return 2.0 * n * h * w * 9 * (c * 16 + 16)

    def launch(self, lib, stream):
        n, h, w, c = self.x.shape
        assert h == 8 and w == 8 and self.x.is_whole()
        P = _scaled(n, self.x.graph)
        rc = lib.kfn_oflow_tail(self.x.ptr, self.k6.ptr, self.b6.ptr if self.b6 is not None else None, self.kp.ptr,
                                self.bpred.ptr if self.bpred is not None else None, self.flow.ptr,
                                self.logits.ptr if self.logits is not None else None, P, c, 16, stream)
        _lib.check(rc, 'kfn_oflow_tail')


def pack_oflow_head_kernel(w):
    """TF HWIO [3,3,32,32] (OFlowNet conv1a) -> the per-lane fragments [144][64] of kfn_oflow_head: fragment
    t = (tap*8 + j)*2 + nb of lane (kq = lane // 16, n = lane % 16) is w[tap][kq*8 + j][nb*16 + n]."""
    w = np.asarray(w, np.float32)
    assert w.shape == (3, 3, 32, 32)
    wt = w.reshape(9, 4, 8, 2, 16)                     # [tap][kq][j][nb][n]
    return np.ascontiguousarray(wt.transpose(0, 2, 3, 1, 4).reshape(144, 64))


def pack_oflow_head_kernel_f16(w):
    """conv1a for kfn_oflow_head_f16: [36][64][4] halfs -- fragment t = (tap*2 + s)*2 + nb of lane (kq, n) holds
    w[tap][kq*8 + 4s + j][nb*16 + n], j = 0..3 (the B operand of one v_mfma_f32_16x16x16_f16)."""
    w = np.asarray(w, np.float32)
    assert w.shape == (3, 3, 32, 32)
    wt = w.reshape(9, 4, 2, 4, 2, 16)                  # [tap][kq][s][j][nb][n]
    return np.ascontiguousarray(wt.transpose(0, 2, 4, 1, 5, 3).reshape(36, 64, 4)).astype(np.float16)


def pack_oflow_upconv_kernel(w):
    """TF conv2d_transpose kernel [3,3,Cout=16,Cin=32] (OFlowNet upconv0) -> the per-lane fragments [72][64] of
    kfn_oflow_tail2: fragment t = tap*8 + j of lane (kq, n) is w[tap][n][kq*8 + j]."""
    w = np.asarray(w, np.float32)
    assert w.shape == (3, 3, 16, 32)
    wt = w.reshape(9, 16, 4, 8)                        # [tap][n][kq][j]
    return np.ascontiguousarray(wt.transpose(0, 3, 2, 1).reshape(72, 64))


def pack_oflow_upconv_kernel_f16(w):
    """upconv0 for kfn_oflow_tail2_f16: [18][64][4] halfs -- fragment t = tap*2 + s of lane (kq, n) holds
    w[tap][n][kq*8 + 4s + j], j = 0..3."""
    w = np.asarray(w, np.float32)
    assert w.shape == (3, 3, 16, 32)
    wt = w.reshape(9, 16, 4, 2, 4)                     # [tap][n][kq][s][j]
    return np.ascontiguousarray(wt.transpose(0, 3, 2, 1, 4).reshape(18, 64, 4)).astype(np.float16)


class OFlowHeadOp(Op):
    """conv0 (from the factored cost-volume maps) + conv1a in one window-resident launch (kfn_oflow_head)."""

    def __init__(self, t, gp, relu0, k1, b1, y, cin, operands_f16=False):
        self.name = 'oflow_head[conv0+conv1a]'
        self.t, self.gp, self.relu0, self.k1, self.b1, self.y, self.cin = t, gp, relu0, k1, b1, y, cin
        self.operands_f16 = operands_f16      # k1 packed by pack_oflow_head_kernel_f16 -> kfn_oflow_head_f16 (config 5)

    def kernel_name(self, lib):
        return 'oflow_head_kernel<%s>' % ('true' if self.operands_f16 else 'false')

    def flops(self):
        """Nominal FLOPs of the two layers this launch completes: conv0 on every window cell + conv1a."""
        n, h, w, _ = self.t.shape
        return 2.0 * n * h * w * (64 * 9 * self.cin * 32 + 16 * 9 * 32 * 32)

    def mfma_flops(self):
        n, h, w, _ = self.t.shape
        return 2.0 * n * h * w * 16 * 9 * 32 * 32

    def launch(self, lib, stream):
        n, h, w, c9 = self.t.shape
        n = _scaled(n, self.t.graph)
        assert c9 == 288 and self.t.ld == c9 and self.gp.ld == c9 and self.y.is_whole() and self.y.shape[1:] == (4, 4, 32)
        fn = lib.kfn_oflow_head_f16 if self.operands_f16 else lib.kfn_oflow_head
        _lib.check(fn(self.t.ptr, self.gp.ptr, n, h, w, int(self.relu0), self.k1.ptr,
                      self.b1.ptr if self.b1 is not None else None, self.y.ptr, stream),
                   'kfn_oflow_head_f16' if self.operands_f16 else 'kfn_oflow_head')


class OFlowTail2Op(Op):
    """upconv0 + conv0 (recomputed from the maps) + conv6 + 'prediction' + softmax + soft-argmax in one
    window-resident launch (kfn_oflow_tail2); neither concat0 nor upconv0's output exist in memory."""

    def __init__(self, t, gp, relu0, x5, ku, bu, k6, b6, kp, bpred, flow, logits=None, operands_f16=False):
        self.name = 'oflow_tail2[upconv0+conv6+prediction+softargmax]'
        self.t, self.gp, self.relu0, self.x5 = t, gp, relu0, x5
        self.operands_f16 = operands_f16      # ku / k6 packed by the *_f16 packers -> kfn_oflow_tail2_f16 (config 5)
        self.ku, self.bu, self.k6, self.b6, self.kp, self.bpred, self.flow, self.logits = ku, bu, k6, b6, kp, bpred, flow, logits

    def kernel_name(self, lib):
        return 'oflow_tail2_kernel<%s>' % ('true' if self.operands_f16 else 'false')

    def flops(self):
        """Nominal (dense, 9-tap) FLOPs of upconv0 (counted per INPUT cell like every transposed conv),
        conv6 and 'prediction' per window."""
        n, h, w, _ = self.t.shape
        return 2.0 * n * h * w * (16 * 9 * 32 * 16 + 64 * 9 * (48 * 16 + 16))

    def mfma_flops(self):
        n, h, w, _ = self.t.shape
        return 2.0 * n * h * w * (9 * 16 * 32 * 16 + 64 * 9 * 48 * 16)

    def launch(self, lib, stream):
        n, h, w, c9 = self.t.shape
        n = _scaled(n, self.t.graph)
        assert c9 == 288 and self.t.ld == c9 and self.gp.ld == c9 and self.x5.is_whole() and self.x5.shape[1:] == (4, 4, 32)
        fn = lib.kfn_oflow_tail2_f16 if self.operands_f16 else lib.kfn_oflow_tail2
        _lib.check(fn(self.t.ptr, self.gp.ptr, n, h, w, int(self.relu0), self.x5.ptr, self.ku.ptr,
                      self.bu.ptr if self.bu is not None else None, self.k6.ptr,
                      self.b6.ptr if self.b6 is not None else None, self.kp.ptr,
                      self.bpred.ptr if self.bpred is not None else None, self.flow.ptr,
                      self.logits.ptr if self.logits is not None else None, stream),
                   'kfn_oflow_tail2_f16' if self.operands_f16 else 'kfn_oflow_tail2')


class CopyChannelsOp(Op):
    def __init__(self, src, dst):
        self.name = 'copy_channels'
        self.src, self.dst = src, dst

    def launch(self, lib, stream):
        rc = lib.kfn_copy_channels(self.src.ptr, self.src.ld, self.dst.ptr, self.dst.ld, self.src.pixels,
                                   self.src.C, stream)
        _lib.check(rc, 'kfn_copy_channels')


class MemcpyOp(Op):
    """Device-to-device copy of a channel-dense tensor (feature ring hand-over)."""

    def __init__(self, src, dst):
        self.name = 'memcpy_d2d'
        self.src, self.dst = src, dst

    def launch(self, lib, stream):
        n, h, w, c = self.src.shape
        assert self.src.ld == c and self.dst.ld == c and self.dst.shape == self.src.shape
        rc = lib.kfn_memcpy_d2d(self.dst.ptr, self.src.ptr, n * h * w * c * 4, stream)
        _lib.check(rc, 'kfn_memcpy_d2d')


class ApplyTransformOp(Op):
    """KFNet/util.py:12-40 as its own launch (kfn_apply_transform): y = (T [x;1])[0:3].  `transform`: host 4x4 or [B,4,4]."""

    def __init__(self, x, y, transform):
        self.name = 'apply_transform'
        self.x, self.y = x, y
        T = np.ascontiguousarray(np.asarray(transform, dtype=np.float32))
        if T.shape not in ((4, 4), (x.shape[0], 4, 4)):
            raise ValueError('ApplyTransform: transform must be 4x4 or Bx4x4, got %s' % (T.shape,))
        self.transform = T
        self._dev = None

    def launch(self, lib, stream):
        import torch
        if self._dev is None:
            self._dev = torch.from_numpy(self.transform.reshape(-1)).to(self.x.graph.device)
        n, h, w, _ = self.x.shape
        rc = lib.kfn_apply_transform(self.x.ptr, self.x.ld, self._dev.data_ptr(), int(self.transform.ndim == 3),
                                     _scaled(n, self.x.graph), h, w, self.y.ptr, self.y.ld, stream)
        _lib.check(rc, 'kfn_apply_transform')


class PixelMapOp(Op):
    """KFNet/util.py:42-63 (kfn_pixel_map): y[b,i,j] = (j, i), optionally normalised by the intrinsics."""

    def __init__(self, y, normalize=False, u=0.0, v=0.0, focal_x=1.0, focal_y=1.0):
        self.name = 'pixel_map'
        self.y = y
        self.normalize = bool(normalize)
        self.u, self.v, self.fx, self.fy = float(u), float(v), float(focal_x), float(focal_y)

    def launch(self, lib, stream):
        n, h, w, _ = self.y.shape
        rc = lib.kfn_pixel_map(self.y.ptr, self.y.ld, n, h, w, int(self.normalize), self.u, self.v, self.fx, self.fy, stream)
        _lib.check(rc, 'kfn_pixel_map')


class BilinearSamplerOp(Op):
    """tools/util.py:3-94 (kfn_bilinear_sampler): imgs [B,Hs,Ws,C] sampled at coords [B,Ht,Wt,2] (x, y)."""

    def __init__(self, imgs, coords, y):
        self.name = 'bilinear_sampler'
        self.imgs, self.coords, self.y = imgs, coords, y

    def launch(self, lib, stream):
        b, hs, ws, c = self.imgs.shape
        _, ht, wt, _ = self.coords.shape
        rc = lib.kfn_bilinear_sampler(self.imgs.ptr, self.imgs.ld, _scaled(b, self.imgs.graph), hs, ws, c, self.coords.ptr,
                                      self.coords.ld, ht, wt, self.y.ptr, self.y.ld, stream)
        _lib.check(rc, 'kfn_bilinear_sampler')


class KalmanScanOp(Op):
    """S sequences x T frames of warp + Kalman fuse (+NIS, transform, emit)."""

    def __init__(self, flow, sigma_t, meas, state, records, temp=None, nis=None, S=1, T=1, H=0, W=0,
                 reset_period=500, min_uncertainty=1e-5, nis_gate=0.0, transform=None, kf_raw=None,
                 raw_on_reset=False):
        self.name = 'kalman_scan'
        self.flow, self.sigma_t, self.meas, self.state = flow, sigma_t, meas, state
        self.records, self.temp, self.nis = records, temp, nis
        self.kf_raw, self.raw_on_reset = kf_raw, raw_on_reset   # kfn_kalman_scan_ex debug outputs
        self.S, self.T, self.H, self.W = S, T, H, W
        self.reset_period = reset_period
        self.min_uncertainty = min_uncertainty
        self.nis_gate = nis_gate
        self.transform = transform
        self.t0 = 0
        self._scratch = None     # second copy of the state for grids that do not fit the LDS (the caller's to provide)

    def scratch_ptr(self, lib, d):
        need = C.c_size_t(0)
        _lib.check(lib.kfn_kalman_scan_scratch_bytes(C.byref(d), C.byref(need)), 'kfn_kalman_scan_scratch_bytes')
        if need.value == 0:
            return None
        if self._scratch is None or self._scratch.numel() * 4 < need.value:
            import torch
            self._scratch = torch.empty((need.value + 3) // 4, dtype=torch.float32, device=self.state.root_storage.buf.device)
        return self._scratch.data_ptr()

    def launch(self, lib, stream):
        d = _lib.KalmanDesc(S=self.S, T=self.T, H=self.H, W=self.W, t0=int(self.t0),
                            reset_period=int(self.reset_period), min_uncertainty=self.min_uncertainty,
                            nis_gate=float(self.nis_gate), has_transform=int(self.transform is not None))
        if self.transform is not None:
            t = np.asarray(self.transform, dtype=np.float32)[:3, :4].reshape(-1)
            for i in range(12):
                d.transform[i] = float(t[i])
        rc = lib.kfn_kalman_scan_ex(C.byref(d), self.flow.ptr, self.sigma_t.ptr, self.meas.ptr, self.state.ptr,
                                    self.records.ptr, self.temp.ptr if self.temp is not None else None,
                                    self.nis.ptr if self.nis is not None else None,
                                    self.kf_raw.ptr if self.kf_raw is not None else None,
                                    int(bool(self.raw_on_reset)), self.scratch_ptr(lib, d), stream)
        _lib.check(rc, 'kfn_kalman_scan')


class KalmanFuseOp(Op):
    """Stand-alone KFNet.BuildKFCoord (+ optional NIS) on packed [.,.,.,4] tensors."""

    def __init__(self, pred, meas, out, nis=None, symmetric_variance=False):
        self.name = 'kalman_fuse2' if symmetric_variance else 'kalman_fuse'
        self.pred, self.meas, self.out, self.nis = pred, meas, out, nis
        self.symmetric_variance = symmetric_variance     # KFNet.GetKFCoord2's posterior (KFNet/KFNet.py:487-502)

    def launch(self, lib, stream):
        for t in (self.pred, self.meas, self.out):
            assert t.ld == 4 and t.C == 4
        if self.symmetric_variance:
            assert self.nis is None
            _lib.check(lib.kfn_kalman_fuse2(self.pred.ptr, self.meas.ptr, self.out.ptr, self.out.pixels, stream),
                       'kfn_kalman_fuse2')
            return
        rc = lib.kfn_kalman_fuse(self.pred.ptr, self.meas.ptr, self.out.ptr,
                                 self.nis.ptr if self.nis is not None else None, self.out.pixels, stream)
        _lib.check(rc, 'kfn_kalman_fuse')


# ---------------------------------------------------------------------------------------
class Graph(object):
    def __init__(self):
        self.ops = []
        self.storages = []
        self.params = {}
        self.first_conv = {}  # id(img tensor) -> FirstConvOp
        self.device = None
        self.debug_prob = False
        self.fuse_flow_head = True  # OFlowNet prediction conv + softmax + soft-argmax in one kernel
        self.fuse_oflow_tail = True  # ... and conv6 in front of it (kfn_oflow_tail: the 8x8x48 patch stays in LDS)
        self.fuse_cost_volume = True  # BuildCoordVolume generated inside OFlowNet conv0's loader
        # OFlowNet's window-grid ends window-resident (kfn_oflow_head: conv0 + conv1a; kfn_oflow_tail2: upconv0 + conv0 +
        # conv6 + prediction + soft-argmax): conv0's [P,8,8,32] output, concat0 and the gather launch disappear.
        # Needs the factored cost volume and the fused tail.
        self.fuse_oflow_window = True
        # conv_operands == 'f16' only: conv1a / upconv0 / conv6 inside the two window-resident launches on fp16 MFMAs
        # (kfn_oflow_head_f16, kfn_oflow_tail2_f16) like OFlowNet's other layers in that mode
        self.oflow_tail_f16 = True
        # conv_operands == 'f16', fp16 activations in and out, 3x3 stride 1, 64 -> 64 channels (SCoordNet conv1b): the
        # row-streaming kernel with register-resident weights (kfn_conv3x3_c64_f16) instead of the implicit GEMM
        self.conv64_rows_f16 = True
        self.lds_bytes_per_cu = 160 * 1024   # gfx950; KFNetEngine overwrites it with kfn_device_info's answer before the
                                             # graph is built (the window-resident OFlowNet tail needs 136 000 B per workgroup)
        # Winograd F(2x2,3x3) for 3x3 stride-1 convs with at least this many in/out channels
        # (0 disables).  Below ~128 channels the [tiles][16][Cout] workspace traffic outweighs
        # the 2.25x MFMA saving.
        self.winograd_min_channels = 128
        self.winograd_ws = None
        # single-kernel Winograd (kfn_conv2d_winograd_fused: no workspace, no output-transform launch)
        # for 3x3 stride-1 layers with at least winograd_fused_min_channels in/out channels; layers it
        # cannot take (Cin % 16, fewer than 4 tile rows) fall back to the two-kernel form above
        self.winograd_fused = True
        self.winograd_fused_min_channels = 32   # (32: the flow-feature tower's feat3, 0.49 -> 0.35 ms at batch 32)
        # F(4x4,3x3) (kfn_conv2d_winograd_f43) for 3x3 stride-1 layers with at least this many INPUT channels (0 = off):
        # the K loop must amortise the 36-position transforms and the cross-wave output reduction
        self.winograd_f43_min_channels = 64
        self.winograd_f43_min_workgroups = 128   # below this the launch leaves the chip idle: F(2x2,3x3) (Network.conv)
        # split-K for F(4x4,3x3) launches of fewer workgroups than CUs (single frames: conv4b 160, conv5 80, conv6 40): the
        # split WinogradF43ConvOp.best_k_split picks (eight-wave form; 0 / 1 = never split)
        self.winograd_f43_max_k_split = 8
        # kfn_conv_desc.weights_path of the fp16-activation direct kernel (0 = the library's choice; 1 = operands through registers,
        # 2 = weights by LDS-DMA, 3 = both operand tiles by LDS-DMA): an A/B and debugging switch
        self.conv_weights_path = 0
        self.winograd_s2_max_k_split = 8         # the same for the eight-wave stride-2 kernel (conv4a at batch 1: 320 workgroups)
        # the eight-wave form of the F(4x4,3x3) kernel (wino4b_kernel; measured at batch 32 against the four-wave form, same
        # box: conv1b 2.66 -> 2.34 ms, conv2b 6.98 -> 6.54, conv3b 6.43 -> 6.13, conv4b 6.08 -> 5.85)
        self.winograd_f43_eight_wave = True
        # ... and of the stride-2 polyphase kernel (wino_s2b_kernel; batch 32, same box: conv2a 4.65 -> 4.54 ms, conv3a 7.35 ->
        # 7.22, conv4a 7.19 -> 7.09)
        self.winograd_s2_eight_wave = True
        # stride-2 layers through the polyphase + F(4,2) kernel (wino_s2c_kernel: 81 instead of 100 products per 16 outputs, 8-14 %
        # faster per layer at the bench batch) when the launch has at least this many of its 16x16-pixel x 128-channel workgroups
        # (4 rounds of 256 CUs); below that the F(2,2) kernel with its smaller blocks (and split-K) fills the chip better.  0 = never.
        self.winograd_s2_f42 = True
        self.winograd_s2_f42_min_workgroups = 1024
        # dense intermediate tensors between two launches of wino4b_kernel / wino_s2c_kernel live channel-blocked (KFN_LAYOUT_C16:
        # per image [C/16][H][W][16]) -- a Winograd super-step's 16 input channels of a patch row are then contiguous (measured at
        # batch 20: the F(4x4,3x3) kernel -2 ... -3 % per layer).  False = NHWC everywhere.
        self.activation_layout_c16 = True
        self.winograd_fused_max_channels = 1024
        # 3x3 stride-2 layers of even-sized images with at least this many input / 128 output channels take the
        # polyphase F(2,2) kernel (kfn_conv2d_winograd_s2); 0 = always the direct implicit GEMM
        self.winograd_s2_min_channels = 64
        self.winograd_s2_f16 = False
        self.window_fc = True   # 3x3 stride-1 layers on 2x2 images as one dense matrix per window (WindowFcConvOp)   # fp16-operand mode: stride-2 layers stay on the direct fp16 kernel (faster)
        # conv0 of OFlowNet by linearity: per-pixel class convolutions + a gather instead of a
        # 3x3 conv on every one of the 64 window cells (see kfn_cost_volume_gather)
        self.factor_cost_volume = True
        # 'f32': exact fp32 MFMA everywhere (the parity path).  'f16': convolutions with
        # Cin % 32 == 0 round their operands to fp16 while staging (fp32 accumulate, fp32
        # activations in memory) -- BASELINE config 5 "fp16 convs + fp32 Kalman", own tolerance.
        # 'f16x3': forward convs split every operand into hi+lo halfs (3 fp16 MFMAs per product,
        # fp32 accumulate): fp32-class accuracy at the fp16 MFMA rate (experimental, opt-in).
        self.conv_operands = 'f32'
        # 'f16' mode only: convolutions under these variable scopes keep their ACTIVATIONS in fp16 in memory
        # (outputs with >= 64 channels; every consumer there is another convolution).  SCoordNet is 94 % of the
        # FLOPs and of the activation traffic; the flow-feature tower and OFlowNet (cost-volume gather, concat
        # views, patch-resident tail) stay fp32 in memory.  () = fp32 activations everywhere (round 2's form).
        self.f16_activation_scopes = ('ScoreNet',)
        self.f16x3_min_channels = 64
        self.active = (1, 1)  # (frames in this launch, frames the graph was built for)

    def winograd_lds_fits(self, stride, cin, cout, wino_form=0, operand_dtype=_lib.OPERAND_F32):
        """Does the Winograd kernel that would be launched for this 3x3 layer fit this device's LDS?  (kfn_winograd_lds_bytes
        against lds_bytes_per_cu: the F(4x4,3x3) and four-wave forms need 144-157 KB of the 160 KiB a gfx950 CU has; on a
        part with less Network.conv falls through to the next route instead of failing in the first launch.)"""
        d = _lib.ConvDesc(N=1, H=64, W=64, Cin=cin, ldx=cin, Cout=cout, cout_pad=-(-cout // 32) * 32, ldy=cout, kh=3, kw=3,
                          stride=stride, operand_dtype=operand_dtype, wino_form=wino_form)
        nb = C.c_int(0)
        if _lib.load().kfn_winograd_lds_bytes(C.byref(d), C.byref(nb)) != _lib.KFN_OK:
            return False
        return nb.value <= self.lds_bytes_per_cu

    # -- construction -------------------------------------------------------------------
    def placeholder(self, shape, dtype='f32', name=None):
        t = Tensor(self, shape, dtype, name)
        t.external = True
        return t

    def tensor(self, shape, dtype='f32', name=None):
        return Tensor(self, shape, dtype, name)

    def add(self, op):
        self.ops.append(op)
        return op

    def winograd_workspace(self, nbytes):
        """One workspace shared by all Winograd layers (they run one after the other)."""
        if self.winograd_ws is None:
            self.winograd_ws = Storage(0, 'f32')
            self.storages.append(self.winograd_ws)
        self.winograd_ws.numel = max(self.winograd_ws.numel, (nbytes + 3) // 4)
        return self.winograd_ws

    def variable(self, name, shape, pack):
        full = (current_scope() + '/' + name) if current_scope() else name
        if full in self.params:  # tf.AUTO_REUSE
            return self.params[full]
        return Param(self, full, shape, pack)

    def derived_variable(self, source_param, tag, pack):
        """A second device layout of an existing variable (same TF name in the container)."""
        key = source_param.source + '#' + tag
        if key in self.params:
            return self.params[key]
        return Param(self, key, source_param.shape, pack, source=source_param.source)

    # -- execution ----------------------------------------------------------------------
    def finalize(self, device='cuda:0'):
        import torch
        if not torch.cuda.is_available():
            raise _lib.KfnError('no GPU visible: kfnet_amd executes only through libkfnet_hip.so on a '
                                'gfx950 device (there is no CPU fallback)')
        _lib.load()
        self.device = torch.device(device)
        for op in self.ops:          # routing decisions that depend on the final buffer bindings
            if hasattr(op, 'resolve'):
                op.resolve()
        self.assign_layouts()
        for s in self.storages:
            s.allocate(self.device)
        return self

    @staticmethod
    def _tensor_refs(op):
        """Every Tensor an op holds, as (attribute name, tensor) -- through lists / tuples / dicts too."""
        def walk(name, v, depth=0):
            if isinstance(v, Tensor):
                yield name, v
            elif isinstance(v, (list, tuple)) and depth < 3:
                for e in v:
                    for r in walk(name, e, depth + 1):
                        yield r
            elif isinstance(v, dict) and depth < 3:
                for e in v.values():
                    for r in walk(name, e, depth + 1):
                        yield r
        for name, v in vars(op).items():
            for r in walk(name, v):
                yield r

    def assign_layouts(self):
        """Channel-blocked layout for the tensors that only Winograd launches touch (after resolve(): the routes are final).
        A tensor qualifies when it is a dense fp32 root tensor of the graph with C % 16 == 0, EVERY reference any op holds is
        the tensor itself (no views), exactly one op writes it (as .y) and all others read it as .x, and all of them launch a
        kernel that takes KFN_LAYOUT_C16.  Anything else -- inputs, outputs, concat members, tensors the cost volume, the
        heads or the engine's copies read -- stays NHWC.  Returns the names of the blocked tensors."""
        if not self.activation_layout_c16:
            return []
        refs = {}
        for op in self.ops:
            for attr, t in self._tensor_refs(op):
                root = t
                while root.base is not None:
                    root = root.base
                refs.setdefault(id(root), (root, []))[1].append((op, attr, t))
        blocked = []
        for root, uses in refs.values():
            root._layout = 'nhwc'          # (idempotent: a graph that grew since the last call is judged afresh)
            if root.external or not root.is_whole() or root.dtype != 'f32' or root.shape[3] % 16 != 0 or root._slot != 0:
                continue
            if any(t is not root for _, _, t in uses):
                continue
            writers = [op for op, attr, _ in uses if attr == 'y']
            readers = [op for op, attr, _ in uses if attr == 'x']
            if len(writers) != 1 or not readers or len(writers) + len(readers) != len(uses):
                continue
            if not all(isinstance(op, ConvOp) and op.takes_c16() for op in writers + readers):
                continue
            root.set_layout('c16')
            blocked.append(root.name)
        return blocked

    def load_weights(self, W, strict=True):
        """RestoreFromScope analogue (KFNet/train.py:317-321): W is {tf_name: ndarray}."""
        import torch
        if self.device is None:
            raise _lib.KfnError('call Graph.finalize(device) before load_weights')
        for key, p in self.params.items():
            name = p.source
            if name not in W:
                if strict:
                    raise KeyError('weight %s missing from the container' % name)
                continue
            arr = np.asarray(W[name], dtype=np.float32)
            if tuple(arr.shape) != p.shape:
                raise ValueError('weight %s has shape %s, graph wants %s' % (name, arr.shape, p.shape))
            packed = p.pack(arr)
            p.packed_shape = packed.shape
            p.storage = torch.from_numpy(np.ascontiguousarray(packed)).to(self.device)
        return self

    def run(self, stream=None, ops=None, active=None):
        """Launch `ops` (default: all) in order.  active=(nb, B) runs only the first nb of
        the B batch entries the graph was built for."""
        import torch
        lib = _lib.load()
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self.active = active if active is not None else (1, 1)
        try:
            for op in (ops if ops is not None else self.ops):
                op.launch(lib, stream)
        finally:
            self.active = (1, 1)

    def autotune(self, ops=None, reps=2, verbose=False):
        """Pick the fastest tile configuration for every MFMA conv launch by timing all of
        them once on the live buffers (a few seconds at engine start-up).  Layer outputs
        are overwritten with identical results, so this is safe before the first run."""
        import torch
        lib = _lib.load()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        chosen = {}
        for op in (ops if ops is not None else self.ops):
            if not isinstance(op, ConvOp) or op.epilogue == _lib.EPI_L2NORM:
                continue
            best = None
            for cfg in (1, 2, 3, 4, 5, 6, 7):
                op.config = cfg
                try:
                    op.launch(lib, stream)   # warm / validity
                except _lib.KfnError:
                    continue
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    op.launch(lib, stream)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / reps
                if best is None or ms < best[0]:
                    best = (ms, cfg)
            op.config = best[1] if best else _lib.CFG_AUTO
            chosen[op.name + '@%d' % id(op)] = (op.config, best[0] if best else None)
            if verbose:
                print('autotune %-12s -> cfg %d (%.3f ms)' % (op.name, op.config, best[0] if best else -1))
        return chosen

    def total_flops(self):
        return sum(op.flops() for op in self.ops if hasattr(op, 'flops'))
