"""Process-model head: a small U-Net over every pixel's 8x8x32 local cost volume, behind
the reference's `OFlowNet` class surface `(inputs, window_area, is_training, reuse)`,
`setup()`, `GetOutput()` (reference: cnn_wrapper/OFlowNet.py:6-57).

Encoder 8x8 -> 4x4 -> 2x2 -> 1x1, decoder back up with skip concatenations:

    conv0 (32) - conv1a (32, /2) - conv1b (32) - conv2a (64, /2) - conv2b (64)
        - conv3a (128, /2) - conv3b (128) - upconv2 (64, x2)
    concat2 = [upconv2, conv2b] - conv4 (64) - upconv1 (32, x2)
    concat1 = [upconv1, conv1b] - conv5 (32) - upconv0 (16, x2)
    concat0 = [upconv0, conv0 ] - conv6 (16) - prediction (1, linear)
    heads: softmax over the 64 'prediction' cells; fc1 (64) - fc2 (32) - uncertainty (1) on conv3b
"""
from .. import _lib
from ..graph import (ConvOp, FlowHeadOp, FlowOp, OFlowTailOp, as_f16, pack_bias, pack_dense_kernel,
                     pack_flow_head_kernel, pack_oflow_tail_kernel)
from .network import Network

# Program for Network: ('conv'|'deconv', name, channels, stride) or ('join', name, producer, skip)
ENCODER = (('conv', 'conv0', 32, 1), ('conv', 'conv1a', 32, 2), ('conv', 'conv1b', 32, 1),
           ('conv', 'conv2a', 64, 2), ('conv', 'conv2b', 64, 1), ('conv', 'conv3a', 128, 2),
           ('conv', 'conv3b', 128, 1), ('deconv', 'upconv2', 64, 2))
DECODER = (('join', 'concat2', 'upconv2', 'conv2b'), ('conv', 'conv4', 64, 1), ('deconv', 'upconv1', 32, 2),
           ('join', 'concat1', 'upconv1', 'conv1b'), ('conv', 'conv5', 32, 1), ('deconv', 'upconv0', 16, 2),
           ('join', 'concat0', 'upconv0', 'conv0'), ('conv', 'conv6', 16, 1))
LOGITS = 'prediction'
FC = (('fc1', 64, True), ('fc2', 32, True), ('uncertainty', 1, False))


# Input BHWx8x8xC cost volume; output flow probabilities and transition uncertainty
class OFlowNet(Network):
    def __init__(self, inputs, window_area, is_training, reuse=False):
        n, h, w = inputs['input'].get_shape().as_list()[:3]
        self.batch_size, self.height, self.width = n, h, w   # set before setup() runs, as in the reference
        self.window_area = window_area
        Network.__init__(self, inputs, is_training, reuse=reuse)

    def setup(self):
        net = self.feed('input')
        for step in ENCODER + DECODER:
            kind, name = step[0], step[1]
            if kind == 'join':
                net = self.feed(step[2], step[3]).concat(-1, name=name)
            elif kind == 'deconv':
                net = net.deconv(3, step[2], step[3], name=name)
            else:
                net = net.conv(3, step[2], step[3], name=name)
        net.conv(3, 1, 1, relu=False, name=LOGITS)

    def _dense(self, x, units, name, relu, epilogue=_lib.EPI_NONE):
        """Fully connected layer (tf.layers.dense in the reference, OFlowNet.py:50-55) run as a
        1x1 convolution on the [BHW,1,1,C] tensor."""
        g = self.graph
        n, h, w, cin = x.shape
        y = g.tensor((n, h, w, units), name=name)
        f16 = g.conv_operands == 'f16' and cin % 32 == 0
        kern = g.variable(name + '/kernel', (cin, units), as_f16(pack_dense_kernel) if f16 else pack_dense_kernel)
        bias = g.variable(name + '/bias', (units,), pack_bias)
        self._emit(ConvOp(name, x, y, kern, bias, 1, 1, 1, relu, epilogue=epilogue,
                          operand_dtype=_lib.OPERAND_F16 if f16 else _lib.OPERAND_F32))
        self.layers[name] = y
        return y

    def GetOutput(self):
        """Returns (prob_map [BHW,1,1,window_area], uncertainty [BHW,1,1,1]) like reference
        OFlowNet.py:43-57: softmax over the window cells of `prediction`, and
        exp(fc3(fc2(fc1(conv3b)))) * 1e-2.

        The kernel that evaluates the softmax also produces the soft-argmax flow that
        KFNet.BuildOFlowNet derives from the probabilities (reference KFNet.py:381-385); it is
        attached as `prob_map.flow` ([BHW,1,1,2]).  With `graph.fuse_flow_head` the
        `prediction` convolution itself is absorbed too (kfn_flow_head) and the probabilities are
        only materialised when `graph.debug_prob` is set."""
        g = self.graph
        logits = self.get_output_by_name(LOGITS)  # BHW x 8 x 8 x 1
        n, window = logits.shape[0], logits.shape[1]
        assert logits.shape[1] * logits.shape[2] == self.window_area and logits.is_whole()
        prob_map = g.tensor((n, 1, 1, self.window_area), name='prob')
        flow = g.tensor((n, 1, 1, 2), name='flow')
        pred_op = [op for op in self.ops if op.name == LOGITS][0]
        x = pred_op.x
        fusable = (g.fuse_flow_head and not g.debug_prob and window == 8 and x.is_whole()
                   and x.shape[3] % 4 == 0 and x.shape[3] <= 32)
        if fusable:
            # the logits are produced and consumed on chip; the 'prediction' tensor is not written
            g.ops.remove(pred_op)
            self.ops.remove(pred_op)
            kern = g.params[pred_op.kernel.name]
            kern.pack = pack_flow_head_kernel
            c6 = [op for op in self.ops if op.y is x and type(op) is ConvOp]
            tail = (g.fuse_oflow_tail and len(c6) == 1 and c6[0].x.is_whole() and c6[0].x.shape[1:] == (8, 8, 48)
                    and x.shape[3] == 16 and c6[0].kh == 3 and c6[0].stride == 1 and c6[0].relu
                    and c6[0].operand_dtype == _lib.OPERAND_F32 and c6[0].kernel.storage is None)
            if tail:
                # conv6 too: its input patch stays in LDS, its output never exists in memory
                g.ops.remove(c6[0])
                self.ops.remove(c6[0])
                k6 = g.params[c6[0].kernel.name]
                k6.pack = pack_oflow_tail_kernel
                self._emit(OFlowTailOp(c6[0].x, k6, c6[0].bias, kern, pred_op.bias, flow))
            else:
                self._emit(FlowHeadOp(x, kern, pred_op.bias, flow))
        else:
            self._emit(FlowOp(logits, flow, prob_map if g.debug_prob else None, window))
        prob_map.flow = flow
        feat = self.get_output_by_name('conv3b')  # BHW x 1 x 1 x 128
        for name, units, relu in FC:
            last = (name == FC[-1][0])
            feat = self._dense(feat, units, name, relu, epilogue=_lib.EPI_EXP_1E2 if last else _lib.EPI_NONE)
        return prob_map, feat
