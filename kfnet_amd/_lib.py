"""ctypes binding of libkfnet_hip.so (include/kfnet_hip.h).

The product path has NO CPU fallback: if the library is missing, or an entry point
returns an error, a KfnError is raised.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libkfnet_hip.so')

KFN_OK = 0
ABI_VERSION = 10
COMM_ID_BYTES = 128
EPI_NONE, EPI_L2NORM, EPI_EXP_CH3, EPI_EXP_1E2 = 0, 1, 2, 3
OPERAND_F32, OPERAND_F16, OPERAND_F16X3 = 0, 1, 2
ACT_F32, ACT_F16 = 0, 1
LAYOUT_NHWC, LAYOUT_C16 = 0, 1                            # kfn_conv_desc.x_layout / y_layout (KFN_LAYOUT_*)
PNG_OK, PNG_UNSUPPORTED, PNG_ERROR = 0, 1, 2
WINO_ORDER_AUTO, WINO_ORDER_M_FAST, WINO_ORDER_N_FAST = 0, 1, 2
CFG_128x256 = 9
CFG_256x64, CFG_256x256, CFG_256x256_W8 = 12, 13, 14
WINO_FORM_F43_FOUR_WAVE, WINO_FORM_F43_EIGHT_WAVE = 2, 3   # kfn_conv_desc.wino_form for kfn_conv2d_winograd_f43
WINO_FORM_S2_EIGHT_WAVE = 4                                # ... for kfn_conv2d_winograd_s2
WINO_FORM_S2_F42 = 5                                       # ... its polyphase + F(4,2) form (wino_s2c_kernel)
CFG_AUTO, CFG_160x128, CFG_128x128, CFG_128x64, CFG_128x32, CFG_64x64, CFG_256x32, CFG_192x64 = 0, 1, 2, 3, 4, 5, 6, 7


class KfnError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """kfn_conv_desc (include/kfnet_hip.h).  `struct_size` is filled in here: the library copies that many bytes and
    reads every later field as 0, so the struct can grow at its end without breaking older hosts
    (tests/test_host_logic.py::test_conv_desc_matches_header_and_integration_doc keeps the three field lists equal)."""
    _fields_ = [(n, C.c_int32) for n in (
        'struct_size', 'N', 'H', 'W', 'Cin', 'ldx', 'Cout', 'cout_pad', 'ldy', 'kh', 'kw', 'stride',
        'transposed', 'relu', 'epilogue', 'config', 'operand_dtype', 'wino_order', 'wino_form',
        'x_dtype', 'y_dtype', 'k_step', 'weights_path', 'x_layout', 'y_layout')]

    def __init__(self, *args, **kw):
        super(ConvDesc, self).__init__(*args, **kw)
        if not self.struct_size:
            self.struct_size = C.sizeof(ConvDesc)


class KalmanDesc(C.Structure):
    _fields_ = [('S', C.c_int32), ('T', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('t0', C.c_int32), ('reset_period', C.c_int32),
                ('min_uncertainty', C.c_float), ('nis_gate', C.c_float),
                ('has_transform', C.c_int32), ('transform', C.c_float * 12)]


# name -> (restype, argtypes); every symbol declared in include/kfnet_hip.h
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    'kfn_last_error': (C.c_char_p, []),
    'kfn_abi_version': (_i, []),
    'kfn_device_info': (_i, [_i, C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    'kfn_malloc': (_i, [C.POINTER(_vp), _sz]),
    'kfn_free': (_i, [_vp]),
    'kfn_memcpy_h2d': (_i, [_vp, _vp, _sz, _vp]),
    'kfn_memcpy_d2h': (_i, [_vp, _vp, _sz, _vp]),
    'kfn_memcpy_d2d': (_i, [_vp, _vp, _sz, _vp]),
    'kfn_memset': (_i, [_vp, _i, _sz, _vp]),
    'kfn_stream_create': (_i, [C.POINTER(_vp)]),
    'kfn_stream_destroy': (_i, [_vp]),
    'kfn_stream_sync': (_i, [_vp]),
    'kfn_event_create': (_i, [C.POINTER(_vp)]),
    'kfn_event_destroy': (_i, [_vp]),
    'kfn_event_record': (_i, [_vp, _vp]),
    'kfn_event_elapsed_ms': (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    'kfn_conv2d_nhwc': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    'kfn_conv2d_out_shape': (_i, [C.POINTER(ConvDesc), C.POINTER(_i), C.POINTER(_i)]),
    'kfn_conv2d_plan': (_i, [C.POINTER(ConvDesc), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'kfn_winograd_plan': (_i, [C.POINTER(ConvDesc), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'kfn_winograd_workspace_bytes': (_i, [C.POINTER(ConvDesc), C.POINTER(_sz)]),
    'kfn_conv2d_winograd': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'kfn_conv2d_winograd_fused': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    'kfn_winograd_fused_supported': (_i, [C.POINTER(ConvDesc)]),
    'kfn_conv2d_winograd_s2': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    'kfn_winograd_s2_supported': (_i, [C.POINTER(ConvDesc)]),
    'kfn_winograd_s2_splitk_workspace_bytes': (_i, [C.POINTER(ConvDesc), _i, C.POINTER(_sz)]),
    'kfn_conv2d_winograd_s2_splitk': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'kfn_conv2d_winograd_f43': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    'kfn_winograd_f43_supported': (_i, [C.POINTER(ConvDesc)]),
    'kfn_winograd_lds_bytes': (_i, [C.POINTER(ConvDesc), C.POINTER(_i)]),
    'kfn_winograd_f43_splitk_workspace_bytes': (_i, [C.POINTER(ConvDesc), _i, C.POINTER(_sz)]),
    'kfn_conv2d_winograd_f43_splitk': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'kfn_conv3x3_c64_f16': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    'kfn_conv3x3_c64_f16_supported': (_i, [C.POINTER(ConvDesc)]),
    'kfn_first_conv_u8': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    'kfn_first_conv_u8_ex': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp]),
    'kfn_cost_volume': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'kfn_cost_volume_conv': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'kfn_pad_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'kfn_cost_volume_gather': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'kfn_flow_softargmax': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'kfn_flow_head': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'kfn_oflow_tail': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'kfn_oflow_head': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'kfn_oflow_head_f16': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'kfn_oflow_tail2': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'kfn_oflow_tail2_f16': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'kfn_kalman_scan_scratch_bytes': (_i, [C.POINTER(KalmanDesc), C.POINTER(_sz)]),
    'kfn_kalman_scan': (_i, [C.POINTER(KalmanDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'kfn_kalman_scan_ex': (_i, [C.POINTER(KalmanDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'kfn_eval_metrics': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.c_float, C.c_float, _vp, _vp, _vp]),
    'kfn_kalman_fuse': (_i, [_vp, _vp, _vp, _vp, C.c_long, _vp]),
    'kfn_kalman_arith_probe': (_i, [_vp, _vp, _vp, C.c_long, _vp]),
    'kfn_kalman_fuse2': (_i, [_vp, _vp, _vp, C.c_long, _vp]),
    'kfn_copy_channels': (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    'kfn_apply_transform': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    'kfn_pixel_map': (_i, [_vp, _i, _i, _i, _i, _i, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    'kfn_bilinear_sampler': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp]),
    'kfn_decode_png_rgb8': (_i, [C.POINTER(C.c_char_p), _i, _i, _i, _vp, C.POINTER(_i), _i]),
    'kfn_comm_available': (_i, []),
    'kfn_comm_unique_id': (_i, [_vp, _sz]),
    'kfn_comm_init': (_i, [C.POINTER(_vp), _i, _i, _vp, _i]),
    'kfn_comm_destroy': (_i, [_vp]),
    'kfn_comm_rank': (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    'kfn_send_state': (_i, [_vp, _i, _vp, _i, _i, _vp]),
    'kfn_recv_state': (_i, [_vp, _i, _vp, _i, _i, _vp]),
}

_lib = None


def load():
    """Load the HIP library (once).  Fails loudly: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KfnError('%s not found -- run `python -m kfnet_amd.build` (hipcc, gfx950); '
                       'kfnet_amd has no CPU fallback' % LIB_PATH)
    try:
        # torch bundles its own libamdhip64; the library must bind to that runtime, not to a second copy from
        # /opt/rocm, or torch's device pointers mean nothing to it ("no ROCm-capable device is detected")
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.kfn_abi_version() != ABI_VERSION:
        raise KfnError('libkfnet_hip.so ABI version mismatch')
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != KFN_OK:
        msg = load().kfn_last_error()
        raise KfnError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))
