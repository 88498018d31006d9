"""Graph-level helpers with the reference's names and argument meaning (KFNet/util.py:5-63).

On eval.py's path these are fused into the scan kernel (csrc/kfn_kalman.hip); a script that calls them itself -- the
reference's eval.py:57-59 applies `ApplyTransform` to three coordinate maps at Python level -- gets the same arithmetic as a
graph op over its own small launch (csrc/kfn_util_ops.hip).  Tensors are kfnet_amd.graph.Tensor handles; the op is appended to
the tensor's graph and runs with it (Graph.run).  The augmentation helpers of KFNet/util.py:66-136 are training-only and not
part of this package."""
import numpy as np

from ..graph import ApplyTransformOp, PixelMapOp, Tensor


def HomoCoord(coords):
    """KFNet/util.py:5-10 (append a channel of ones) exists in the reference only as ApplyTransform's first step; here the
    translation column is added inside kfn_apply_transform and the homogeneous tensor is never materialised."""
    raise NotImplementedError('HomoCoord is folded into ApplyTransform (kfn_apply_transform); it has no launch of its own')


def ApplyTransform(coords, transform, inverse=False, name=None):
    """KFNet/util.py:12-40.  coords: Tensor [B,H,W,3] (a channel view of a packed buffer is fine); transform: 4x4 or
    Bx4x4 array (host; the reference passes a tf.constant built from np.loadtxt, KFNet/eval.py:51); inverse=True applies
    inv(transform) (tf.matrix_inverse -> np.linalg.inv in fp32 like the graph op).  Returns a new Tensor [B,H,W,3]."""
    if not isinstance(coords, Tensor):
        raise TypeError('ApplyTransform: coords must be a kfnet_amd graph Tensor')
    shape = coords.get_shape().as_list()
    if len(shape) != 4 or shape[3] != 3:
        raise ValueError('ApplyTransform: coords must be BxHxWx3, got %s' % (shape,))
    T = np.asarray(transform, dtype=np.float32)
    if inverse:
        T = np.linalg.inv(T).astype(np.float32)
    g = coords.graph
    y = g.tensor(shape, name=name or 'apply_transform')
    g.add(ApplyTransformOp(coords, y, T))
    return y


def GetPixelMap(batch_size, height, width, normalize=False, spec=None, graph=None, name=None):
    """KFNet/util.py:42-63: [B,H,W,2] with map[b,y,x] = (x, y); normalize=True: ((x - spec.u) / spec.focal_x,
    (y - spec.v) / spec.focal_y).  `graph`: the kfnet_amd Graph the map belongs to (TF's implicit default graph has no
    counterpart here)."""
    if graph is None:
        raise ValueError('GetPixelMap needs graph= (there is no implicit default graph)')
    if normalize and spec is None:
        raise ValueError('GetPixelMap(normalize=True) needs spec (u, v, focal_x, focal_y)')
    y = graph.tensor((batch_size, height, width, 2), name=name or 'pixel_map')
    if normalize:
        graph.add(PixelMapOp(y, True, spec.u, spec.v, spec.focal_x, spec.focal_y))
    else:
        graph.add(PixelMapOp(y))
    return y
