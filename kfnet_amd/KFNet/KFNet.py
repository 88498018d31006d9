"""Model assembly with the reference's `KFNet/KFNet.py` class surface (KFNetDataSpec,
KFNet and its Build*/Get* methods), re-designed for a frame-batched MI355X pipeline.

Differences from the reference's graph, all result-preserving:
  * `images` is a uint8 [B,H,W,3] batch of B CONSECUTIVE frames, not a (t-1, t) pair.
    Both towers run once per frame; the previous batch's last feature map is kept in
    slot 0 of a [B+1,h,w,32] ring so frame pairs are (slot b, slot b+1).  No layer couples
    batch elements, so this equals the reference's batch-2 evaluation of every pair.
  * The recurrent state is one packed [1,h,w,4] tensor (x,y,z,sigma); `last_coord` /
    `last_uncertainty` are channel views of it.
  * warp + Kalman fuse + NIS + transform/emit is one persistent scan launch over the B
    frames (kfn_kalman_scan) instead of ~35 TF elementwise/gather ops per frame.
"""
import numpy as np

from .. import _lib
from ..cnn_wrapper.network import Network
from ..cnn_wrapper.OFlowNet import OFlowNet
from ..cnn_wrapper.SCoordNet import SCoordNet
from ..graph import (ConvOp, CostVolumeConvOp, CostVolumeGatherOp, CostVolumeOp, DerivedConvOp, Graph, KalmanScanOp, MemcpyOp,
                     PadOp, Tensor, as_f16, pack_cvol_bias, pack_cvol_g_kernel, pack_cvol_t_kernel, variable_scope)


class KFNetDataSpec():
    """KFNet/KFNet.py:6-50 (the scene table only feeds training-schedule fields)."""

    def __init__(self,
                 batch_size=4,
                 image_size=(480, 640),
                 channels=3,
                 crop_size=(480, 640),
                 focal_x=525.,
                 focal_y=525.,
                 u=320.,
                 v=240.,
                 scene='stairs'):
        self.batch_size = batch_size
        self.image_size = image_size
        self.channels = channels
        self.crop_size = crop_size
        self.focal_x = focal_x
        self.focal_y = focal_y
        self.u = u
        self.v = v
        self.scene = scene
        self.image_num = 2000
        self.sequence_length = 500
        self.num_sequence = 4
        table = {'chess': (4000, 1000, 4), 'fire': (2000, 1000, 2), 'heads': (1000, 1000, 1),
                 'office': (6000, None, None), 'pumpkin': (4000, None, None),
                 'redkitchen': (7000, None, None), 'stairs': (2000, 500, 4)}
        if scene in table:
            n, sl, ns = table[scene]
            self.image_num = n
            if sl is not None:
                self.sequence_length = sl
                self.num_sequence = ns
        self.stepvalue = self.image_num * 75 // self.batch_size
        self.max_steps = self.stepvalue * 5


class _FeatTower(Network):
    """KFNet.BuildOFlowFeat's 7 convs (KFNet/KFNet.py:318-338) expressed with the DSL."""

    def setup(self):
        from ..cnn_wrapper.network import PreprocessedImage
        img = self.layers['input']
        self.layers['preprocess'] = PreprocessedImage(img, 'preprocess')
        (self.feed('preprocess')
         .conv(3, 16, 1, name='feat1')    # 640x480
         .conv(3, 32, 2, name='feat2')    # 320x240
         .conv(3, 32, 1, name='feat3')    # 320x240
         .conv(3, 64, 2, name='feat4')    # 160x120
         .conv(3, 64, 1, name='feat5')    # 160x120
         .conv(3, 128, 2, name='feat6')   # 80x60
         .conv(3, 32, 1, relu=False, name='feat7'))  # 80x60


class KFNet():
    def __init__(self, images, spec, train_scoordnet=False, train_oflownet=False, dropout_rate=0.5,
                 seed=None, reuse=True):
        if train_scoordnet or train_oflownet:
            raise NotImplementedError('training is out of scope of the MI355X prediction path')
        self.focal_x = spec.focal_x
        self.focal_y = spec.focal_y
        self.u = spec.u
        self.v = spec.v

        self.reuse = reuse
        self.train_scoordnet = train_scoordnet
        self.train_oflownet = train_oflownet
        self.seed = seed
        self.dropout_rate = dropout_rate

        self.graph = images.graph
        shape = images.get_shape().as_list()
        self.batch_size = shape[0]
        self.height = shape[1]
        self.width = shape[2]
        self.flow_sample_rate = 8
        self.min_uncertainty = 1e-5

        self.images = images
        self.frame_ops = []   # image-only stage (towers)
        self.pair_ops = []    # image-pair stage (cost volume, OFlowNet, flow)
        self.scan_ops = []    # recurrent stage
        self.scoordnet = self.BuildSCoordNet()
        self.temp_feat_maps = self.BuildOFlowFeat()
        self._kalman = None

    ####################### I/O #######################
    def GetInputImages(self):
        return self.images

    def GetMeasureCoord(self):
        return self.scoordnet.GetOutput()

    def GetMeasureCoord2(self):
        """KFNet/KFNet.py:470-474 picks batch index 1 of the pair; here every batch element
        is a frame whose measurement is used, so this is GetMeasureCoord."""
        return self.GetMeasureCoord()

    def GetKFCoordRecursive(self, last_coord, last_uncertainty, transform=None, reset_period=0,
                            nis_gate=0.0, emit_temp=False, emit_nis=False):
        """KFNet/KFNet.py:85-100 for all B frames of the batch, in order.

        last_coord / last_uncertainty must be the channel views (0:3, 3:4) of ONE packed
        [1,h,w,4] state tensor; the state is updated in place (this is eval.py's
        SetVariableByName feedback, KFNet/eval.py:96-104).  Returns
        (temp_coord, temp_uncertainty, KF_coord, KF_uncertainty): temp_* are views of the
        optional [B,h,w,4] prediction buffer (None unless emit_temp), KF_* are views of the
        state (the raw KF estimate of the batch's LAST frame).  The per-frame output
        records (T.x, 1/sigma) are `self.records` [B,h,w,4].
        """
        state = last_coord.base
        if state is None or last_uncertainty.base is not state or state.shape[3] != 4:
            raise ValueError('last_coord/last_uncertainty must be channel views of one packed [1,h,w,4] state')
        measure_coord, measure_uncertainty = self.GetMeasureCoord2()
        meas = measure_coord.base  # packed [B,h,w,4]
        B = self.batch_size
        feats = self.temp_feat_maps
        feat_map1 = feats.batch(0, B, name='feat_map1')
        feat_map2 = feats.batch(1, B, name='feat_map2')
        n_ops = len(self.graph.ops)
        flow, transition_uncertainty = self.BuildOFlowNet(feat_map1, feat_map2, last_coord, last_uncertainty)
        self.pair_ops = self.graph.ops[n_ops:]
        _, h, w, _ = meas.shape
        g = self.graph
        self.records = g.tensor((B, h, w, 4), name='records')
        self.temp = g.tensor((B, h, w, 4), name='temp') if emit_temp else None
        self.nis = g.tensor((B, h, w, 3), name='NIS') if emit_nis else None
        op = KalmanScanOp(flow, transition_uncertainty, meas, state, self.records, self.temp, self.nis,
                          S=1, T=B, H=h, W=w, reset_period=reset_period, min_uncertainty=self.min_uncertainty,
                          nis_gate=nis_gate, transform=transform)
        g.add(op)
        self._kalman = op
        # hand the last feature map over to slot 0 for the next batch
        cp = MemcpyOp(feats.batch(B, 1), feats.batch(0, 1))
        g.add(cp)
        self.scan_ops = [op, cp]
        temp_coord = self.temp.channels(0, 3) if emit_temp else None
        temp_unc = self.temp.channels(3, 1) if emit_temp else None
        return temp_coord, temp_unc, state.channels(0, 3), state.channels(3, 1)

    def _fuse_oflow_window(self, tt, gp, conv0, gather_op, cin):
        """OFlowNet's two window-grid ends as window-resident launches (csrc/kfn_oflow_fused.hip):
        [gather -> conv0.y] + conv1a  ->  kfn_oflow_head;   upconv0 + concat0 + oflow_tail  ->  kfn_oflow_tail2.
        conv0's output, upconv0's output and concat0 are then never written.  Applied only when the graph has
        exactly the reference's wiring (cnn_wrapper/OFlowNet.py:19-20,36-41); otherwise nothing changes."""
        from ..graph import (OFlowHeadOp, OFlowTail2Op, OFlowTailOp, pack_bias, pack_oflow_head_kernel,
                             pack_oflow_head_kernel_f16, pack_oflow_tail_kernel_f16, pack_oflow_upconv_kernel,
                             pack_oflow_upconv_kernel_f16)
        g = self.graph
        net_ops = self.oflownet.ops
        if not g.fuse_oflow_window:
            return
        # kfn_oflow_tail2 needs 4 x 32 560 B of dynamic LDS per workgroup: gfx950's 160 KiB.  On a part with less the
        # round-2 launches (gather + conv1a, upconv0 + kfn_oflow_tail) stay in place.
        if getattr(g, 'lds_bytes_per_cu', 160 * 1024) < 4 * 32560:
            return
        y0 = conv0.y
        tails = [op for op in net_ops if isinstance(op, OFlowTailOp)]
        if len(tails) != 1 or tails[0].logits is not None:
            return
        tail = tails[0]
        cat = tail.x                                    # concat0 [P,8,8,48] = [upconv0 | conv0]
        if not (y0.base is None and y0.storage is cat.storage and y0.ch_off == 16 and y0.ld == 48 and y0.shape[3] == 32):
            return
        c1a = [op for op in net_ops if type(op) is ConvOp and op.x is y0]
        up0 = [op for op in net_ops if type(op) is ConvOp and op.transposed and op.y.base is None
               and op.y.storage is cat.storage and op.y.ch_off == 0]
        if len(c1a) != 1 or len(up0) != 1:
            return
        c1a, up0 = c1a[0], up0[0]
        ok = (c1a.kh == 3 and c1a.kw == 3 and c1a.stride == 2 and not c1a.transposed and c1a.relu
              and c1a.epilogue == _lib.EPI_NONE and c1a.y.is_whole() and c1a.y.shape[1:] == (4, 4, 32)
              and c1a.kernel.storage is None and cin == 32
              and up0.kh == 3 and up0.stride == 2 and up0.relu and up0.y.shape[3] == 16 and up0.x.is_whole()
              and up0.x.shape[1:] == (4, 4, 32) and up0.kernel.storage is None and up0.x.dtype == 'f32')
        # nobody else may read the tensors that disappear
        gone = (y0, cat, up0.y)
        for op in g.ops:
            if op in (c1a, up0, tail, gather_op):
                continue
            if any(getattr(op, attr, None) in gone for attr in ('x', 'src', 'logits')):
                ok = False
        if not ok:
            return
        # "fp16 convs" (BASELINE config 5): conv1a, upconv0 and conv6 multiply in fp16 like OFlowNet's other layers in that
        # mode (both launches are MFMA-bound in fp32: 0.37 -> 0.1x ms and 1.29 -> 0.3x ms per 16 frames of 540x960)
        h16 = g.conv_operands == 'f16' and g.oflow_tail_f16
        c1a.kernel.pack = pack_oflow_head_kernel_f16 if h16 else pack_oflow_head_kernel
        up0.kernel.pack = pack_oflow_upconv_kernel_f16 if h16 else pack_oflow_upconv_kernel
        if h16:
            tail.k6.pack = pack_oflow_tail_kernel_f16
        for b in (c1a.bias, up0.bias):
            if b is not None:
                b.pack = pack_bias
        head = OFlowHeadOp(tt, gp, conv0.relu, c1a.kernel, c1a.bias, c1a.y, cin, operands_f16=h16)
        tail2 = OFlowTail2Op(tt, gp, conv0.relu, up0.x, up0.kernel, up0.bias, tail.k6, tail.b6, tail.kp, tail.bpred,
                             tail.flow, operands_f16=h16)
        for lst in (g.ops, net_ops):
            lst.remove(gather_op)
            lst.remove(up0)
            lst[lst.index(c1a)] = head
            lst[lst.index(tail)] = tail2
        if cat.storage in g.storages:
            g.storages.remove(cat.storage)

    def BuildKFCoord(self, last_coord, last_uncertainty, measure_coord, measure_uncertainty):
        """KFNet/KFNet.py:148-162 as a stand-alone launch (kfn_kalman_fuse) on packed
        [.,.,.,4] tensors; inside GetKFCoordRecursive it is fused into the scan kernel."""
        from ..graph import KalmanFuseOp
        pred, meas = last_coord.base, measure_coord.base
        if pred is None or meas is None or last_uncertainty.base is not pred or measure_uncertainty.base is not meas:
            raise ValueError('BuildKFCoord wants channel views of packed (coord, sigma) tensors')
        out = self.graph.tensor(pred.shape, name='KF')
        self.graph.add(KalmanFuseOp(pred, meas, out))
        return out.channels(0, 3), out.channels(3, 1)

    def GetTemporalCoord2(self, coord_map1=None, uncertainty1=None):
        """KFNet/KFNet.py:476-485: the process-model prediction (x^-, sigma^-) of every frame of the batch.  Here it
        is an output of the scan kernel: build GetKFCoordRecursive(..., emit_temp=True) first (its last_coord /
        last_uncertainty ARE coord_map1 / uncertainty1); returns channel views of the [B,h,w,4] prediction buffer."""
        if getattr(self, 'temp', None) is None:
            raise ValueError('build GetKFCoordRecursive(..., emit_temp=True) first')
        return self.temp.channels(0, 3), self.temp.channels(3, 1)

    def GetKFCoord2(self, KF_coord_map1=None, KF_uncertainty1=None):
        """KFNet/KFNet.py:487-502: the fusion with the symmetric-form posterior variance (1-K)^2 P^- + K^2 R, as a
        stand-alone launch (kfn_kalman_fuse2) on the prediction and measurement buffers of the batch.  Like the
        reference's it is not part of the recursive eval path (which uses BuildKFCoord); for a batch of one frame --
        the reference's case, one step per sess.run -- it is exactly GetKFCoord2 of the state fed to
        GetKFCoordRecursive."""
        from ..graph import KalmanFuseOp
        temp_coord, temp_unc = self.GetTemporalCoord2()
        meas = self.GetMeasureCoord2()[0].base
        out = self.graph.tensor(self.temp.shape, name='KF2')
        op = KalmanFuseOp(self.temp, meas, out, symmetric_variance=True)
        self.graph.add(op)
        self.scan_ops.append(op)
        return out.channels(0, 3), out.channels(3, 1)

    def GetNIS(self, measure_coord_map, measure_uncertainty_map, temp_coord_map, temp_uncertainty_map):
        """KFNet/KFNet.py:164-184; produced by the scan kernel when emit_nis is set."""
        if getattr(self, 'nis', None) is None:
            raise ValueError('build GetKFCoordRecursive(..., emit_nis=True) first')
        return self.nis
    ####################### eof I/O #######################

    def BuildSCoordNet(self):
        n0 = len(self.graph.ops)
        with variable_scope('ScoreNet'):
            net = SCoordNet({'input': self.images},
                            is_training=self.train_scoordnet,
                            focal_x=self.focal_x,
                            focal_y=self.focal_y,
                            u=self.u,
                            v=self.v,
                            dropout_rate=self.dropout_rate,
                            seed=self.seed,
                            reuse=self.reuse)
        self.frame_ops += self.graph.ops[n0:]
        return net

    def BuildOFlowFeat(self):
        """KFNet/KFNet.py:315-341.  Returns the [B+1,h,w,32] feature ring; slots 1..B are
        written by feat7 (+ fused l2_normalize), slot 0 is the previous batch's last map."""
        g = self.graph
        n0 = len(g.ops)
        with variable_scope('Temporal'):
            tower = _FeatTower({'input': self.images}, is_training=False)
        feat7 = tower.get_output_by_name('feat7')
        B, h, w, c = feat7.shape
        ring = g.tensor((B + 1, h, w, c), name='feat_ring')
        # re-home feat7's output into slots 1..B of the ring and fuse the L2 normalisation
        g.storages.remove(feat7.storage)
        feat7.base = ring
        feat7.rel_off = 0
        feat7.rel_batch = 1
        tower.set_epilogue('feat7', _lib.EPI_L2NORM)     # tf.nn.l2_normalize (KFNet/KFNet.py:340)
        self.feat_tower = tower
        self.frame_ops += [op for op in g.ops[n0:] if op not in self.frame_ops]
        return ring

    def BuildCoordVolume(self, feat_map1, feat_map2, window_size):
        """KFNet/KFNet.py:343-359 (+ the reshape of :372): returns the [BHW,w,w,C] volume
        tensor and the [w*w,2] (x,y) offsets (host ndarray)."""
        n, h, w, c = feat_map2.shape
        vol = self.graph.tensor((n * h * w, window_size, window_size, c), name='diff_feat')
        self.graph.add(CostVolumeOp(feat_map1, feat_map2, vol, window_size))
        half = window_size // 2
        offs = np.array([(j - half, i - half) for i in range(window_size) for j in range(window_size)],
                        dtype=np.float32)
        return vol, offs

    def BuildOFlowNet(self, feat_map1, feat_map2, coord_map1, uncertainty1):
        """KFNet/KFNet.py:361-403, image-pair half: cost volume -> OFlowNet -> soft-argmax
        flow and transition sigma.  The state-dependent half (bilinear warp of
        coord_map1/uncertainty1 and variance propagation, :386-401) runs inside the scan
        kernel, so this returns (flow [BHW,1,1,2], transition_uncertainty [BHW,1,1,1])."""
        window_size = 64 // self.flow_sample_rate  # ensure a flow field 96x96 of original resolution
        window_area = window_size * window_size
        diff_feats, shift_offsets = self.BuildCoordVolume(feat_map1, feat_map2, window_size)
        with variable_scope('Temporal'):
            coord_flow_net = OFlowNet({'input': diff_feats}, window_area, is_training=self.train_oflownet,
                                      reuse=self.reuse)
            prob, transition_uncertainty = coord_flow_net.GetOutput()
        self.oflownet = coord_flow_net
        self.prob = prob
        self._fuse_cost_volume(diff_feats, feat_map1, feat_map2, window_size)
        return prob.flow, transition_uncertainty

    def _fuse_cost_volume(self, vol, feat_map1, feat_map2, window_size):
        """Replace (cost_volume launch, conv0 launch reading the volume) by ONE launch that
        generates the volume in conv0's loader; the [BHW,8,8,C] tensor is then never allocated."""
        g = self.graph
        if not g.fuse_cost_volume or window_size != 8:
            return
        cv = [op for op in g.ops if isinstance(op, CostVolumeOp) and op.vol is vol]
        c0 = [op for op in g.ops if isinstance(op, ConvOp) and op.x is vol]
        if len(cv) != 1 or len(c0) != 1 or c0[0].kh != 3 or c0[0].stride != 1 or c0[0].transposed:
            return
        conv0 = c0[0]
        net_ops = self.oflownet.ops
        if (conv0.operand_dtype == _lib.OPERAND_F16 and g.factor_cost_volume and conv0.bias is not None
                and conv0.kernel.storage is None):
            # fp16-operand mode (BASELINE config 5): OFlowNet's window-grid ends run on the fp32 window-resident
            # kernels all the same, which need the factored form of conv0 (two class convolutions, 15 GFLOP / batch;
            # THEY round their operands to fp16 like every other convolution of this mode -- see `od` below)
            conv0.operand_dtype = _lib.OPERAND_F32
        if conv0.operand_dtype != _lib.OPERAND_F32:
            return   # the loader-generated volume exists for the fp32 kernel only
        if g.factor_cost_volume and conv0.bias is not None:
            # conv0 is linear and V[p,cell] = f2[p] - f1[p+cell-4]: two per-pixel convolutions
            # with the 9 border-class kernels + one gather replace the per-cell 3x3 conv
            n, h, w, c = feat_map2.shape
            co = conv0.y.shape[3]
            f1p = g.tensor((n, h + 4, w + 4, c), name='feat_map1_padded')
            gp = g.tensor((n, h + 4, w + 4, 9 * co), name='conv0_G')
            tt = g.tensor((n, h, w, 9 * co), name='conv0_T')
            # fp16-operand mode: the two class convolutions round their operands like every other convolution there
            # (-0.4 ms per 16-frame batch).  f2 and f1 are then rounded separately and T - G is formed afterwards, so the
            # error is relative to |f| (unit-norm features: 2^-11) rather than to |f2 - f1|, and identical features no
            # longer cancel exactly (T uses fp16(sum of taps), G the per-tap fp16 weights); the flows of the two paths
            # agree to 2.9e-3 px (config 5's tolerance test asserts < 0.05 px), DESIGN 4 (config 5).
            h16 = g.conv_operands == 'f16' and c % 32 == 0
            od = _lib.OPERAND_F16 if h16 else _lib.OPERAND_F32
            wg = g.derived_variable(conv0.kernel, 'cvol_G', as_f16(pack_cvol_g_kernel) if h16 else pack_cvol_g_kernel)
            wt = g.derived_variable(conv0.kernel, 'cvol_T', as_f16(pack_cvol_t_kernel) if h16 else pack_cvol_t_kernel)
            b9 = g.derived_variable(conv0.bias, 'cvol_b9', pack_cvol_bias)
            new_ops = [PadOp(feat_map1, f1p, 2),
                       DerivedConvOp('conv0[G]', f1p, gp, wg, None, 3, 3, 1, False, operand_dtype=od),
                       DerivedConvOp('conv0[T]', feat_map2, tt, wt, b9, 1, 1, 1, False, operand_dtype=od),
                       CostVolumeGatherOp(tt, gp, conv0.y, conv0.relu, c)]
            i = g.ops.index(conv0)
            g.ops[i:i + 1] = new_ops
            g.ops.remove(cv[0])
            j = net_ops.index(conv0)
            net_ops[j:j + 1] = new_ops
            # the TF-layout copies conv0 itself would have used are not needed
            for prm in (conv0.kernel, conv0.bias):
                if not any(getattr(op, 'kernel', None) is prm or getattr(op, 'bias', None) is prm for op in g.ops):
                    g.params.pop(prm.name, None)
            if vol.storage in g.storages:
                g.storages.remove(vol.storage)
            self._fuse_oflow_window(tt, gp, conv0, new_ops[3], c)
            return
        # the loader-generated volume feeds the DIRECT kernel: conv0 may have been built as a Winograd layer (3x3
        # stride 1, 32 channels, 8x8 grid), whose kernel variable is packed [Cin/8][16][Cout][8] -- reset the packers
        # to the [cout_pad][9 Cin] layout kfn_cost_volume_conv reads (silently wrong flow otherwise)
        from ..graph import pack_bias, pack_conv_kernel
        if conv0.kernel.storage is not None:
            return   # already uploaded in another layout: keep the unfused pair of launches
        conv0.kernel.pack = pack_conv_kernel
        if conv0.bias is not None:
            conv0.bias.pack = pack_bias
        fused = CostVolumeConvOp(feat_map1, feat_map2, conv0.y, conv0.kernel, conv0.bias, conv0.relu, window_size)
        g.ops[g.ops.index(conv0)] = fused
        g.ops.remove(cv[0])
        net_ops[net_ops.index(conv0)] = fused
        if vol.storage in g.storages:
            g.storages.remove(vol.storage)
