"""SCoordNet, the measurement network -- same class surface as the reference's
cnn_wrapper/SCoordNet.py:4-44, building kfnet_amd.graph launches instead of TF ops."""
from .. import _lib
from .network import Network, PreprocessedImage, layer


class SCoordNet(Network):
    def __init__(self, inputs, is_training, focal_x, focal_y, u, v, dropout_rate=0.5, seed=None, reuse=False):
        Network.__init__(self, inputs, is_training, dropout_rate, seed, reuse)
        self.focal_x = focal_x
        self.focal_y = focal_y
        self.u = u
        self.v = v

        images = inputs['input']
        shape = images.get_shape().as_list()
        self.batch_size = shape[0]
        self.height = shape[1]
        self.width = shape[2]

    def setup(self):
        (self.feed('input')
         .preprocess(name='preprocess')
         .conv(3, 64, 1, name='conv1a')
         .conv(3, 64, 1, name='conv1b')
         .conv(3, 256, 2, name='conv2a')
         .conv(3, 256, 1, name='conv2b')
         .conv(3, 512, 2, name='conv3a')
         .conv(3, 512, 1, name='conv3b')
         .conv(3, 1024, 2, name='conv4a')
         .conv(3, 1024, 1, name='conv4b')
         .conv(3, 512, 1, name='conv5')
         .conv(3, 256, 1, name='conv6')
         .conv(1, 128, 1, name='conv7')
         .conv(1, 4, 1, relu=False, name='prediction'))

    @layer
    def preprocess(self, input, name):
        """(x - 128) * 0.00625 (SCoordNet.py:34-37); fused into the first conv kernel,
        which reads the uint8 image directly."""
        if input.dtype != 'u8' or input.shape[3] != 3:
            raise TypeError('SCoordNet input must be a uint8 [B,H,W,3] image tensor')
        return PreprocessedImage(input, name)

    def GetOutput(self):
        """coord = prediction[..., 0:3], uncertainty = exp(prediction[..., 3:4])
        (SCoordNet.py:39-44).  The exp is fused into the `prediction` conv's epilogue, so
        after this call channel 3 of the 'prediction' buffer holds exp(raw); both results
        are zero-copy channel views of that [B,h,w,4] buffer."""
        prediction = self.get_output_by_name('prediction')
        for op in self.ops:
            if op.name == 'prediction':
                op.epilogue = _lib.EPI_EXP_CH3
        coord_map = prediction.channels(0, 3, name='coord')
        uncertainty_map = prediction.channels(3, 1, name='uncertainty')
        return coord_map, uncertainty_map
