"""tools/util.py of the reference: `bilinear_sampler` as a graph op (csrc/kfn_util_ops.hip).

On eval.py's path the sampler is fused into the scan kernel (the state lives in LDS there); this is the stand-alone form
with the same arithmetic, for callers that use it at Python level.  `meshgrid` / `flow_warp` (tools/util.py:96-135) are
not used by KFNet and are not part of this package."""
from ..graph import BilinearSamplerOp, Tensor


def bilinear_sampler(imgs, coords, name=None):
    """tools/util.py:3-94.  imgs: Tensor [B,Hs,Ws,C]; coords: Tensor [B,Ht,Wt,2] = (x, y) in source pixels.  Corner
    indices are clamped to the image and the WEIGHTS are computed from the clamped corners (:55-63), so a sample with
    x < 0 or x >= Ws-1 (y likewise) evaluates to 0 -- x = Ws-1 included; the four products are summed in tf.add_n's order
    (:88-93).  Returns a new Tensor [B,Ht,Wt,C]."""
    if not isinstance(imgs, Tensor) or not isinstance(coords, Tensor):
        raise TypeError('bilinear_sampler: imgs and coords must be kfnet_amd graph Tensors')
    bi, _, _, c = imgs.get_shape().as_list()
    bc, ht, wt, two = coords.get_shape().as_list()
    if two != 2 or bi != bc:
        raise ValueError('bilinear_sampler: imgs %s / coords %s' % (imgs.get_shape().as_list(), coords.get_shape().as_list()))
    if imgs.graph is not coords.graph:
        raise ValueError('bilinear_sampler: imgs and coords belong to different graphs')
    g = imgs.graph
    y = g.tensor((bc, ht, wt, c), name=name or 'bilinear_sampler')
    g.add(BilinearSamplerOp(imgs, coords, y))
    return y
