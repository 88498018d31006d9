"""OFlowNet, the process-model head over the local cost volume -- same class surface as
the reference's cnn_wrapper/OFlowNet.py:6-57."""
from .. import _lib
from ..graph import ConvOp, FlowHeadOp, FlowOp, as_f16, pack_bias, pack_dense_kernel, pack_flow_head_kernel
from .network import Network


# Input BxHxWxN cost volume
# Outout BxHxWx3 coordinate map & BxHxWx1 uncertainty map
class OFlowNet(Network):
    def __init__(self, inputs, window_area, is_training, reuse=False):
        images = inputs['input']
        shape = images.get_shape().as_list()
        self.batch_size = shape[0]
        self.height = shape[1]
        self.width = shape[2]
        self.window_area = window_area

        Network.__init__(self, inputs, is_training, reuse=reuse)

    def setup(self):
        (self.feed('input')
         .conv(3, 32, 1, name='conv0')  # 8x8x32
         .conv(3, 32, 2, name='conv1a') # 4x4x32
         .conv(3, 32, 1, name='conv1b') # 4x4x32
         .conv(3, 64, 2, name='conv2a') # 2x2x64
         .conv(3, 64, 1, name='conv2b') # 2x2x64
         .conv(3, 128, 2, name='conv3a')  # 1x1x128
         .conv(3, 128, 1, name='conv3b')  # 1x1x128
         .deconv(3, 64, 2, name='upconv2'))    # 2x2x64

        (self.feed('upconv2', 'conv2b')
         .concat(-1, name='concat2')    # 2x2x128
         .conv(3, 64, 1, name='conv4')  # 2x2x64
         .deconv(3, 32, 2, name='upconv1')) # 4x4x32

        (self.feed('upconv1', 'conv1b')
         .concat(-1, name='concat1')    # 4x4x64
         .conv(3, 32, 1, name='conv5')  # 4x4x32
         .deconv(3, 16, 2, name='upconv0'))  # 8x8x16

        (self.feed('upconv0', 'conv0')
         .concat(-1, name='concat0')    # 8x8x48
         .conv(3, 16, 1, name='conv6')   # 8x8x16
         .conv(3, 1, 1, relu=False, name='prediction')) # 8x8x1

    def _dense(self, x, units, name, relu, epilogue=_lib.EPI_NONE):
        """tf.layers.dense (OFlowNet.py:50-55) == 1x1 conv on a [BHW,1,1,C] tensor."""
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
        """prob = softmax over the window cells of `prediction`; uncertainty =
        exp(fc3(fc2(fc1(conv3b)))) * 1e-2  (OFlowNet.py:43-57).

        Returns (prob_map [BHW,1,1,window_area], uncertainty [BHW,1,1,1]).  The same kernel
        that evaluates the softmax also produces the soft-argmax flow that
        KFNet.BuildOFlowNet derives from prob (KFNet/KFNet.py:381-385); it is attached as
        `prob_map.flow` ([BHW,1,1,2]) so the caller does not need a second pass.  The
        probabilities themselves are only written when `graph.debug_prob` is set."""
        g = self.graph
        output = self.get_output_by_name('prediction')  # BHWx8x8x1
        n = output.shape[0]
        window = output.shape[1]
        assert output.shape[1] * output.shape[2] == self.window_area and output.is_whole()
        prob_map = g.tensor((n, 1, 1, self.window_area), name='prob')
        flow = g.tensor((n, 1, 1, 2), name='flow')
        pred_op = [op for op in self.ops if op.name == 'prediction'][0]
        x = pred_op.x
        if (g.fuse_flow_head and not g.debug_prob and window == 8 and x.is_whole() and x.shape[3] % 4 == 0
                and x.shape[3] <= 32):
            # the 'prediction' conv is absorbed into the flow head: its 64 logits per pixel are
            # produced and consumed on chip (the 'prediction' layer tensor is not written)
            g.ops.remove(pred_op)
            self.ops.remove(pred_op)
            kern = g.params[pred_op.kernel.name]
            kern.pack = pack_flow_head_kernel
            self._emit(FlowHeadOp(x, kern, pred_op.bias, flow))
        else:
            self._emit(FlowOp(output, flow, prob_map if g.debug_prob else None, window))
        prob_map.flow = flow
        feat = self.get_output_by_name('conv3b')  # BHWx1x1x128
        fc1 = self._dense(feat, 64, 'fc1', True)
        fc2 = self._dense(fc1, 32, 'fc2', True)
        uncertainty = self._dense(fc2, 1, 'uncertainty', False, epilogue=_lib.EPI_EXP_1E2)
        return prob_map, uncertainty
