"""Chainable layer DSL with the reference's `cnn_wrapper.network.Network` surface
(cnn_wrapper/network.py:8-31, 34-437), re-targeted from "add a TF op to the graph" to
"append a libkfnet_hip.so launch to a kfnet_amd.graph.Graph".

Same mechanics: the `@layer` decorator pops `self.terminals`, calls the op, stores the
result in `self.layers[name]`, re-feeds it and returns `self`; `feed` resolves layer
names; `get_unique_name` auto-numbers.  Hot-path layers (`conv`, `deconv`, `concat`) are
backed by HIP kernels.  The reference's other 27 layer methods are never called by
SCoordNet / OFlowNet / KFNet; they keep their names and signatures and raise
NotImplementedError (there is deliberately no CPU/eager fallback in the product path).
"""
import numpy as np

from .. import _lib
from ..graph import (Conv64RowsF16Op, ConvOp, CopyChannelsOp, FirstConvOp, pack_conv64_rows_kernel, Storage, Tensor, WinogradConvOp, WinogradFusedConvOp,
                     WinogradS2ConvOp, WinogradF43ConvOp, WindowFcConvOp, as_f16, pack_winograd_f43_kernel, pack_winograd_f43_kernel_b, pack_winograd_s2_kernel_b, pack_winograd_s2_kernel_c,
                     as_f16x3, pack_bias, pack_conv_kernel, pack_deconv_kernel, pack_first_kernel,
                     pack_bias_x4, pack_window_fc_kernel, pack_winograd_fused_kernel, pack_winograd_kernel,
                     pack_winograd_s2_kernel, current_scope, pack_conv_kernel_chunked)

# Zero padding in default. 'VALID' gives no padding.
DEFAULT_PADDING = 'SAME'

try:
    string_types = (basestring,)  # noqa: F821  (py2 spelling used by the reference)
except NameError:
    string_types = (str,)


def layer(op):
    """Makes a method chainable the way the reference's decorator does (network.py:8-31):
    the wrapped method consumes the current terminal tensor(s), its result is registered in
    `self.layers` under the (possibly auto-generated) `name` keyword, becomes the new
    terminal, and the network itself is returned so that calls can be chained."""

    def chained(net, *args, **kwargs):
        if 'name' not in kwargs:
            kwargs['name'] = net.get_unique_name(op.__name__)
        label = kwargs['name']
        fed = net.terminals
        if len(fed) == 0:
            raise RuntimeError('No input variables found for layer %s.' % label)
        result = op(net, fed[0] if len(fed) == 1 else list(fed), *args, **kwargs)
        net.layers[label] = result
        net.feed(result)
        return net

    chained.__name__ = op.__name__
    chained.__doc__ = op.__doc__
    return chained


class PreprocessedImage(object):
    """Result of `(uint8 image - 128) * 0.00625` that has not been materialised: the
    first convolution consumes it fused (kfn_first_conv_u8)."""

    def __init__(self, source, name):
        self.source = source
        self.name = name
        self.shape = source.shape
        self.graph = source.graph

    def get_shape(self):
        return self.source.get_shape()


def _same_out(n, s):
    return -(-n // s)


def _off_path(name):
    def fn(self, *a, **k):
        raise NotImplementedError(
            "Network.%s is not on the KFNet prediction path (SURVEY.md §2.1) and has no HIP "
            "kernel; only conv / deconv / concat (+ SCoordNet.preprocess) are implemented" % name)
    fn.__name__ = name
    return layer(fn)


class Network(object):
    """Base class of the layer DSL; subclasses describe their architecture in `setup()`."""

    def __init__(self, inputs, is_training, dropout_rate=0.5, seed=None, reuse=False):
        self.inputs = inputs              # name -> input tensor
        self.layers = dict(inputs)        # name -> tensor of every layer built so far
        self.terminals = []               # tensor(s) the next layer call will consume
        self.trainable = self.training = is_training   # kept for signature parity (inference only here)
        self.reuse = reuse
        self.seed = seed
        self.dropout_rate = dropout_rate
        self.ops = []                     # launches appended by this network, in order
        self.setup()

    def setup(self):
        raise NotImplementedError('Must be implemented by the subclass.')

    # ------------------------------------------------------------------------------
    @property
    def graph(self):
        for tensor in self.inputs.values():
            return tensor.graph
        raise RuntimeError('network has no inputs')

    def load(self, data_path, session=None, ignore_missing=False):
        """Weights from the reference's numpy dict format {op_name: {param_name: array}}
        (network.py:60-75).  `session` is accepted for signature compatibility; the arrays go
        to the graph's device buffers under the variable scope active at call time.  The reference's rule: every entry
        of the file is assigned to the variable <scope>/<op_name>/<param_name>; an entry WITHOUT such a variable raises
        unless `ignore_missing` -- variables the file does not mention keep what they hold (a per-scope file therefore
        loads into a graph that also holds the other scope's networks, KFNet/eval.py:66-68)."""
        from ..graph import current_scope
        from ..weights import from_network_load_dict
        table = np.load(data_path, allow_pickle=True).item()
        flat = from_network_load_dict(table, current_scope())
        g = self.graph
        known = set(p.source for p in g.params.values())
        unknown = sorted(k for k in flat if k not in known)
        if unknown and not ignore_missing:
            raise ValueError('Network.load: no variable for %s (ignore_missing=False)' % ', '.join(unknown[:4]))
        g.load_weights(dict((k, v) for k, v in flat.items() if k in known), strict=False)

    def feed(self, *args):
        """Select the input(s) of the next layer call: layer names or tensors."""
        if not args:
            raise AssertionError('feed() needs at least one layer')
        picked = []
        for item in args:
            if isinstance(item, string_types):
                if item not in self.layers:
                    raise KeyError('Unknown layer name fed: %s' % item)
                item = self.layers[item]
            picked.append(item)
        self.terminals = picked
        return self

    def get_output(self):
        """The most recent terminal."""
        return self.terminals[-1]

    def get_output_by_name(self, layer_name):
        return self.layers[layer_name]

    def get_unique_name(self, prefix):
        """`<prefix>_<n>` with n = 1 + number of existing layers whose name starts with prefix."""
        taken = [label for label in self.layers if label.startswith(prefix)]
        return '%s_%d' % (prefix, len(taken) + 1)

    def change_inputs(self, inputs):
        if len(inputs) != 1:
            raise AssertionError('change_inputs() takes exactly one input')
        self.layers.update(inputs)

    def _emit(self, op):
        self.graph.add(op)
        self.ops.append(op)
        return op

    def set_epilogue(self, layer_name, epilogue):
        """Fuse a head operation (KFN_EPI_*: l2_normalize, exp on channel 3, exp * 1e-2) into the
        launch that produces `layer_name`.  Only the direct MFMA convolution carries these
        epilogues; a layer that was routed to the Winograd path is re-routed to the direct
        kernel here (its weights are then packed for that kernel), so that a routing heuristic
        can never silently drop e.g. tf.nn.l2_normalize (KFNet/KFNet.py:340)."""
        from ..graph import pack_conv_kernel as _direct_pack
        hits = [op for op in self.ops if op.name == layer_name and isinstance(op, ConvOp)]
        if len(hits) != 1:
            raise KeyError('set_epilogue: %d convolution launches are named %r' % (len(hits), layer_name))
        op = hits[0]
        if isinstance(op, (WinogradConvOp, WinogradFusedConvOp, WinogradS2ConvOp, WinogradF43ConvOp, Conv64RowsF16Op)):
            if op.kernel.storage is not None:
                raise RuntimeError('set_epilogue(%r): weights are already packed for the %s kernel'
                                   % (layer_name, type(op).__name__))
            # in place (the op object may already sit in other launch lists): same tensors and
            # variables, direct-kernel weight layout.  The fp16-operand forms keep their operand type: the direct
            # kernel reads fp16 activations with chunk-major weights (Network.conv's own rule for that case).
            if op.operand_dtype == _lib.OPERAND_F16:
                chunked = op.x.dtype == 'f16' or op.y.dtype == 'f16'
                op.kernel.pack = as_f16(pack_conv_kernel_chunked if chunked else _direct_pack)
            else:
                op.kernel.pack = _direct_pack
            op.__class__ = ConvOp
            op.workspace = None
            for attr in ('eight_wave',):        # routing state of the class the op just left (desc() no longer reads it)
                op.__dict__.pop(attr, None)
            op.epilogue = epilogue
            return op
        if op.operand_dtype == _lib.OPERAND_F32 and op.y.shape[3] > 32 and epilogue == _lib.EPI_L2NORM:
            raise ValueError('the l2_normalize epilogue needs <= 32 output channels (one MFMA column block)')
        op.epilogue = epilogue
        return op

    def _act_dtype(self, filters):
        """Element type of a convolution's output in memory.  fp16 only in the fp16-operand mode (BASELINE config
        5), under the variable scopes Graph.f16_activation_scopes names, for outputs of >= 64 channels (the narrow
        heads -- 'prediction', 4 channels -- stay fp32)."""
        g = self.graph
        if g.conv_operands != 'f16' or filters < 64 or filters % 8 != 0:
            return 'f32'
        scope = current_scope()
        return 'f16' if any(scope == sc or scope.startswith(sc + '/') for sc in g.f16_activation_scopes) else 'f32'

    # ---- hot-path layers -----------------------------------------------------------
    @layer
    def conv(self, input, kernel_size, filters, strides, name, relu=True, padding=DEFAULT_PADDING,
             biased=True):
        """tf.layers.conv2d (cnn_wrapper/network.py:116-135) -> kfn_conv2d_nhwc, or the
        fused uint8 first layer when fed by `preprocess`."""
        if padding != 'SAME':
            raise NotImplementedError("only padding='SAME' is used by the KFNet path")
        g = self.graph
        k = int(kernel_size)
        if isinstance(input, PreprocessedImage):
            img = input.source
            n, h, w, cin = img.shape
            if not (k == 3 and strides == 1 and relu and cin == 3 and filters % 16 == 0):
                raise NotImplementedError('fused first layer supports 3x3 stride-1 ReLU convs on 3 channels')
            y = g.tensor((n, h, w, filters), dtype=self._act_dtype(filters) if filters == 64 else 'f32', name=name)
            kern = g.variable(name + '/kernel', (3, 3, 3, filters), pack_first_kernel)
            bias = g.variable(name + '/bias', (filters,), pack_bias)
            op = g.first_conv.get(id(img))
            if op is None or len(op.heads) >= 2:
                op = FirstConvOp(img)
                g.first_conv[id(img)] = op
                self._emit(op)
            else:
                self.ops.append(op)
            op.add_head(name, y, kern, bias)
            return y
        n, h, w, cin = input.shape
        f16 = g.conv_operands == 'f16' and cin % 32 == 0
        if input.dtype == 'f16' and not f16:
            raise NotImplementedError('%s: an fp16 activation can only feed an fp16-operand convolution' % name)
        y = g.tensor((n, _same_out(h, strides), _same_out(w, strides), filters),
                     dtype=self._act_dtype(filters) if f16 else 'f32', name=name)
        bias = g.variable(name + '/bias', (filters,), pack_bias) if biased else None
        wmin = g.winograd_min_channels
        # (measured, 16 frames: conv3b 2.07 vs 2.31 ms direct fp16, conv4b 1.81 vs 2.59, conv5 1.00 vs 1.25; at
        #  Cin = 256 the direct kernel is as fast -- conv2b 2.56 vs 2.48 -- and stays)
        if (f16 and input.dtype == 'f32' and y.dtype == 'f32' and k == 3 and strides == 1 and g.winograd_fused and cin >= 512
                and WinogradFusedConvOp.supported(input.shape, cin, filters, _lib.OPERAND_F16)):
            # BASELINE config 5: the four-wave Winograd kernel on fp16 MFMAs (transform in fp32, V and U rounded to fp16)
            kern = g.variable(name + '/kernel', (k, k, cin, filters), as_f16(pack_winograd_fused_kernel))
            self._emit(WinogradFusedConvOp(name, input, y, kern, bias, relu, operand_dtype=_lib.OPERAND_F16))
            return y
        # (the fp16 instantiation of the polyphase stride-2 kernel exists and is tested, but at fp16 MFMA rates it is
        #  latency-bound and the direct fp16 kernel is faster: conv3a 1.20 vs 1.68 ms -- Graph.winograd_s2_f16 = False)
        if (f16 and input.dtype == 'f32' and y.dtype == 'f32' and k == 3 and strides == 2 and g.winograd_s2_f16 and cin >= 64
                and filters >= 128 and WinogradS2ConvOp.supported(input.shape, cin, filters)):
            kern = g.variable(name + '/kernel', (k, k, cin, filters), as_f16(pack_winograd_s2_kernel))
            self._emit(WinogradS2ConvOp(name, input, y, kern, bias, relu, operand_dtype=_lib.OPERAND_F16))
            return y
        if (f16 and k == 3 and strides == 1 and h == 2 and w == 2 and g.window_fc and biased and (4 * cin) % 32 == 0
                and filters % 8 == 0 and input.dtype == 'f32' and y.dtype == 'f32'):
            # OFlowNet's 2x2 level in the fp16-operand mode: the same dense window matrix, operands rounded to fp16
            kern = g.variable(name + '/kernel', (k, k, cin, filters), as_f16(pack_window_fc_kernel))
            bias.pack = pack_bias_x4
            op = WindowFcConvOp(name, input, y, kern, bias, relu)
            op.operand_dtype = _lib.OPERAND_F16
            self._emit(op)
            return y
        if (f16 and g.conv64_rows_f16 and k == 3 and strides == 1 and biased
                and Conv64RowsF16Op.supported(input, y, cin, filters)):
            # (measured, 16 frames of 540x960: 1.05 ms on the 256x64 implicit-GEMM tile, LDS-bound at 0.23 of the fp16 peak)
            kern = g.variable(name + '/kernel', (k, k, cin, filters), pack_conv64_rows_kernel)
            self._emit(Conv64RowsF16Op(name, input, y, kern, bias, relu))
            return y
        if f16:
            # fp16 activations on either side: the tap-innermost kernels with chunk-major weights
            pk = pack_conv_kernel_chunked if (input.dtype == 'f16' or y.dtype == 'f16') else pack_conv_kernel
            kern = g.variable(name + '/kernel', (k, k, cin, filters), as_f16(pk))
            self._emit(ConvOp(name, input, y, kern, bias, k, k, strides, relu, operand_dtype=_lib.OPERAND_F16))
            return y
        if g.conv_operands == 'f16x3' and cin % 32 == 0 and cin >= g.f16x3_min_channels:
            kern = g.variable(name + '/kernel', (k, k, cin, filters), as_f16x3(pack_conv_kernel))
            self._emit(ConvOp(name, input, y, kern, bias, k, k, strides, relu, operand_dtype=_lib.OPERAND_F16X3))
            return y
        if (k == 3 and strides == 1 and h == 2 and w == 2 and g.window_fc and biased and cin % 8 == 0 and filters % 8 == 0
                and g.conv_operands == 'f32'):
            # OFlowNet's 2x2 level: dense window matrix (16 Cin Cout products per window instead of 36)
            kern = g.variable(name + '/kernel', (k, k, cin, filters), pack_window_fc_kernel)
            bias.pack = pack_bias_x4
            self._emit(WindowFcConvOp(name, input, y, kern, bias, relu))
            return y
        f43 = g.winograd_f43_min_channels
        # (measured at batch 32, F(2x2,3x3) -> F(4x4,3x3), profiles/r04_wino4_microbench.log: conv2b 10.02 -> 6.99 ms,
        #  conv3b 9.30 -> 6.41, conv4b 9.18 -> 6.26, conv5 4.61 -> 3.28, conv6 1.24 -> 0.85, conv1b 3.66 -> 2.77, feat5 0.23 -> 0.18;
        #  with 32 output channels -- feat3 -- half of the workgroup idles: 0.31 -> 0.47, stays on wino2_kernel)
        fmin = g.winograd_fused_min_channels
        fused_ok = (k == 3 and strides == 1 and g.winograd_fused and fmin and cin >= fmin and filters >= fmin
                    and cin <= g.winograd_fused_max_channels
                    and min(h, w) >= 8 and WinogradFusedConvOp.supported(input.shape, cin, filters)
                    and g.winograd_lds_fits(1, cin, filters))
        # A launch of fewer than winograd_f43_min_workgroups workgroups (of 32 tiles x 64 channels, one per CU) leaves
        # most of the 256 CUs idle and the smaller F(2x2,3x3) workgroups win -- single frames only (batch 1, F(4x4) ->
        # F(2x2): conv5 80 workgroups 0.287 -> 0.238 ms, conv6 40: 0.151 -> 0.123, feat5 40: 0.031 -> 0.022; from 160
        # workgroups up F(4x4) is ahead: conv4b at batch 1 0.288 against 0.479; profiles/r04_wino4_microbench.log, r4z)
        e8 = bool(g.winograd_f43_eight_wave)
        wgs = WinogradF43ConvOp.workgroups((n, h, w, cin), filters)
        ksplit = 1
        if e8 and g.winograd_f43_max_k_split > 1:
            ksplit = WinogradF43ConvOp.best_k_split(wgs, cin, n * h * w * filters * 4, g.winograd_f43_max_k_split)
        f43_fills = (not fused_ok or wgs * ksplit >= g.winograd_f43_min_workgroups)
        if (k == 3 and strides == 1 and g.winograd_fused and f43 and cin >= f43 and filters >= f43 and f43_fills
                and WinogradF43ConvOp.supported(input.shape, cin, filters, input.ld, y.ld, y.ch_off)
                and g.winograd_lds_fits(1, cin, filters, _lib.WINO_FORM_F43_EIGHT_WAVE if e8 else _lib.WINO_FORM_F43_FOUR_WAVE)):
            kern = g.variable(name + '/kernel', (k, k, cin, filters), pack_winograd_f43_kernel_b if e8 else pack_winograd_f43_kernel)
            op = WinogradF43ConvOp(name, input, y, kern, bias, relu, eight_wave=e8, k_split=ksplit)
            if ksplit > 1:
                # a PRIVATE workspace: the two towers run on two streams, a shared one would race (single frames only:
                # 29-59 MB per split layer at batch 1)
                op.workspace = Storage((op.workspace_bytes() + 3) // 4, 'f32')
                g.storages.append(op.workspace)
            self._emit(op)
            return y
        # (measured, 16-frame batch, single-kernel vs two-kernel form: conv1b 2.3 ms vs 3.3 direct, conv2b 5.60 vs
        #  7.10, conv3b 5.05 vs 5.99, conv4b 4.93 vs 5.42, conv5 2.49 vs 2.73, conv6 0.63 vs 0.77)
        if fused_ok:
            kern = g.variable(name + '/kernel', (k, k, cin, filters), pack_winograd_fused_kernel)
            self._emit(WinogradFusedConvOp(name, input, y, kern, bias, relu))
            return y
        smin = g.winograd_s2_min_channels
        # 3x3 stride-2 layers (SCoordNet conv2a / conv3a / conv4a): polyphase + F(2,2), 25/36 of the direct MFMAs
        e8 = bool(g.winograd_s2_eight_wave)
        if (k == 3 and strides == 2 and smin and cin >= smin and filters >= 128
                and WinogradS2ConvOp.supported(input.shape, cin, filters)
                and g.winograd_lds_fits(2, cin, filters, _lib.WINO_FORM_S2_EIGHT_WAVE if e8 else 0)):
            if (g.winograd_s2_f42 and WinogradS2ConvOp.f42_supported(input.shape, cin, filters)
                    and WinogradS2ConvOp.f42_workgroups(y.shape) >= g.winograd_s2_f42_min_workgroups
                    and g.winograd_lds_fits(2, cin, filters, _lib.WINO_FORM_S2_F42)):
                kern = g.variable(name + '/kernel', (k, k, cin, filters), pack_winograd_s2_kernel_c)
                self._emit(WinogradS2ConvOp(name, input, y, kern, bias, relu, f42=True))
                return y
            kern = g.variable(name + '/kernel', (k, k, cin, filters), pack_winograd_s2_kernel_b if e8 else pack_winograd_s2_kernel)
            ksplit = 1
            if e8 and g.winograd_s2_max_k_split > 1 and filters % 4 == 0:
                ksplit = WinogradS2ConvOp.best_k_split(WinogradS2ConvOp.base_workgroups(y.shape), cin,
                                                       y.shape[0] * y.shape[1] * y.shape[2] * filters * 4, g.winograd_s2_max_k_split)
            op = WinogradS2ConvOp(name, input, y, kern, bias, relu, eight_wave=e8, k_split=ksplit)
            if ksplit > 1:
                op.workspace = Storage((op.workspace_bytes() + 3) // 4, 'f32')     # private: see the F(4x4) split above
                g.storages.append(op.workspace)
            self._emit(op)
            return y
        if (k == 3 and strides == 1 and wmin and cin >= wmin and filters >= wmin and filters % 4 == 0
                and min(h, w) >= 8):
            kern = g.variable(name + '/kernel', (k, k, cin, filters), pack_winograd_kernel)
            op = WinogradConvOp(name, input, y, kern, bias, relu, None)
            op.workspace = g.winograd_workspace(op.workspace_bytes())
            self._emit(op)
            return y
        kern = g.variable(name + '/kernel', (k, k, cin, filters), pack_conv_kernel)
        self._emit(ConvOp(name, input, y, kern, bias, k, k, strides, relu))
        return y

    @layer
    def deconv(self, input, kernel_size, filters, strides, name, relu=True, padding=DEFAULT_PADDING,
               biased=True):
        """tf.layers.conv2d_transpose (cnn_wrapper/network.py:418-437)."""
        if padding != 'SAME' or strides != 2:
            raise NotImplementedError("only stride-2 'SAME' transposed convs are used by the KFNet path")
        g = self.graph
        k = int(kernel_size)
        if getattr(input, 'dtype', 'f32') != 'f32':
            raise NotImplementedError('%s: transposed convolutions read fp32 activations only (an fp16 tensor can only feed '
                                      'Network.conv in the fp16-operand mode)' % name)
        n, h, w, cin = input.shape
        y = g.tensor((n, h * strides, w * strides, filters), name=name)
        f16 = g.conv_operands == 'f16' and cin % 32 == 0
        kern = g.variable(name + '/kernel', (k, k, filters, cin),
                          as_f16(pack_deconv_kernel) if f16 else pack_deconv_kernel)
        bias = g.variable(name + '/bias', (filters,), pack_bias) if biased else None
        self._emit(ConvOp(name, input, y, kern, bias, k, k, strides, relu, transposed=True,
                          operand_dtype=_lib.OPERAND_F16 if f16 else _lib.OPERAND_F32))
        return y

    @layer
    def concat(self, inputs, axis, name):
        """tf.concat on the channel axis (cnn_wrapper/network.py:316-318).  Producers that
        still own a private buffer are re-bound to write straight into their channel
        window of the concat buffer (no copy); anything else is copied."""
        if axis not in (-1, 3):
            raise NotImplementedError('concat is only used on the channel axis')
        g = self.graph
        if any(getattr(t, 'dtype', 'f32') != 'f32' for t in inputs):
            raise NotImplementedError('%s: concat is implemented for fp32 activations only' % name)
        n, h, w, _ = inputs[0].shape
        ctot = sum(t.shape[3] for t in inputs)
        out = g.tensor((n, h, w, ctot), name=name)
        off = 0
        for t in inputs:
            assert t.shape[:3] == (n, h, w)
            c = t.shape[3]
            if isinstance(t, Tensor) and t.is_whole() and not t.external and c % 4 == 0 and off % 4 == 0:
                t.rebind(out.storage, off, ctot)
            else:
                self._emit(CopyChannelsOp(t, out.channels(off, c)))
            off += c
        return out

    # ---- off-path layers: names/signatures kept, no kernels ------------------------
    conv_bn = _off_path('conv_bn')
    conv3d = _off_path('conv3d')
    conv3d_bn = _off_path('conv3d_bn')
    deconv3d = _off_path('deconv3d')
    deconv3d_bn = _off_path('deconv3d_bn')
    relu = _off_path('relu')
    max_pool = _off_path('max_pool')
    avg_pool = _off_path('avg_pool')
    l2_pool = _off_path('l2_pool')
    lrn = _off_path('lrn')
    add = _off_path('add')
    fc = _off_path('fc')
    softmax = _off_path('softmax')
    batch_normalization = _off_path('batch_normalization')
    dropout = _off_path('dropout')
    l2norm = _off_path('l2norm')
    squeeze = _off_path('squeeze')
    maximum = _off_path('maximum')
    tanh = _off_path('tanh')
    reshape = _off_path('reshape')
    slice = _off_path('slice')
    add_n = _off_path('add_n')
    conv_tanh = _off_path('conv_tanh')
