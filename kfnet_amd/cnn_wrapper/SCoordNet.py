"""Measurement network (scene-coordinate regression) behind the reference's `SCoordNet`
class surface: constructor `(inputs, is_training, focal_x, focal_y, u, v, dropout_rate, seed,
reuse)`, a `preprocess` layer, `setup()` and `GetOutput()`
(reference: cnn_wrapper/SCoordNet.py:4-44).

The architecture is held as a table and replayed through the `Network` DSL, so each entry
becomes one `kfnet_amd.graph` launch record instead of a TF op:

    name        k  Cout stride     name        k  Cout stride
    conv1a      3    64   1        conv4a      3  1024   2
    conv1b      3    64   1        conv4b      3  1024   1
    conv2a      3   256   2        conv5       3   512   1
    conv2b      3   256   1        conv6       3   256   1
    conv3a      3   512   2        conv7       1   128   1
    conv3b      3   512   1        prediction  1     4   1  (linear)
"""
from .. import _lib
from .network import Network, PreprocessedImage, layer

# (layer name, kernel size, output channels, stride); every layer but the last has a ReLU
BACKBONE = (
    ('conv1a', 3, 64, 1), ('conv1b', 3, 64, 1),
    ('conv2a', 3, 256, 2), ('conv2b', 3, 256, 1),
    ('conv3a', 3, 512, 2), ('conv3b', 3, 512, 1),
    ('conv4a', 3, 1024, 2), ('conv4b', 3, 1024, 1),
    ('conv5', 3, 512, 1), ('conv6', 3, 256, 1),
    ('conv7', 1, 128, 1),
)
HEAD = ('prediction', 1, 4, 1)   # 3 scene coordinates + log-uncertainty, no activation


class SCoordNet(Network):
    def __init__(self, inputs, is_training, focal_x, focal_y, u, v, dropout_rate=0.5, seed=None, reuse=False):
        # like the reference, the graph is built (setup) before the camera fields are stored
        Network.__init__(self, inputs, is_training, dropout_rate, seed, reuse)
        self.focal_x, self.focal_y, self.u, self.v = focal_x, focal_y, u, v
        n, h, w = inputs['input'].get_shape().as_list()[:3]
        self.batch_size, self.height, self.width = n, h, w

    def setup(self):
        net = self.feed('input').preprocess(name='preprocess')
        for name, ksize, channels, stride in BACKBONE:
            net = net.conv(ksize, channels, stride, name=name)
        name, ksize, channels, stride = HEAD
        net.conv(ksize, channels, stride, relu=False, name=name)

    @layer
    def preprocess(self, input, name):
        """Image normalisation `(x - 128) * 0.00625` (reference SCoordNet.py:34-37).  Nothing is
        launched here: the first convolution kernel reads the uint8 image and normalises on
        the fly (kfn_first_conv_u8)."""
        if input.dtype != 'u8' or input.shape[3] != 3:
            raise TypeError('SCoordNet input must be a uint8 [B,H,W,3] image tensor')
        return PreprocessedImage(input, name)

    def GetOutput(self):
        """(coord_map [B,h,w,3], uncertainty_map [B,h,w,1] = exp(channel 3)) as in reference
        SCoordNet.py:39-44.  The exponential is folded into the `prediction` launch's
        epilogue, so after this call channel 3 of the 'prediction' buffer already holds sigma;
        both results are zero-copy channel views of that buffer."""
        head_name = HEAD[0]
        self.set_epilogue(head_name, _lib.EPI_EXP_CH3)
        packed = self.get_output_by_name(head_name)
        return packed.channels(0, 3, name='coord'), packed.channels(3, 1, name='uncertainty')
