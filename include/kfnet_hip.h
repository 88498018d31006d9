/* libkfnet_hip.so -- C ABI of the MI355X-native (gfx950) KFNet prediction path.
 *
 * The reference (zlthinker/KFNet) has no native/FFI boundary: its hot path is a TF-1.x
 * graph built by the Python `cnn_wrapper.Network` DSL and executed by `sess.run`
 * (KFNet/eval.py:78-83).  Each entry point below replaces the TensorFlow op(s) that one
 * reference call site dispatches to; the Python host in kfnet_amd/ keeps the reference's
 * class/method surface and calls these through ctypes.
 *
 * Conventions
 *   - extern "C"; every function returns KFN_OK (0) or a negative KFN_ERR_*; the text
 *     of the last error on the calling thread is kfn_last_error().  No exceptions cross.
 *   - All tensors are device pointers to fp32, NHWC, unless the name says otherwise.
 *     The caller owns every buffer; the library never allocates or frees tensor memory
 *     behind the caller's back and keeps no per-call state (re-entrant; any number of
 *     host threads may call with distinct streams, on any device: launches go to the
 *     calling thread's current device, kernel attributes are set per device).
 *   - `stream` is a hipStream_t (NULL = the default stream).  All work is asynchronous
 *     on that stream and hipGraph-capturable: no entry point allocates, frees or synchronises
 *     (kfn_malloc / kfn_free / kfn_stream_sync / kfn_event_elapsed_ms / kfn_comm_init are the
 *     plumbing exceptions and say so).  Scratch memory, where a launch needs it, is the caller's
 *     (kfn_winograd_workspace_bytes, kfn_kalman_scan_scratch_bytes).
 *   - A "pixel stride" (ldx / ldy) is the distance in floats between consecutive
 *     pixels; it lets a layer read or write a channel slice of a wider buffer, which is
 *     how `Network.concat` (cnn_wrapper/network.py:316-318) is realised without copies.
 */
#ifndef KFNET_HIP_H_
#define KFNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KFN_OK 0
#define KFN_ERR_ARG (-1)
#define KFN_ERR_HIP (-2)
#define KFN_ERR_UNSUPPORTED (-3)

/* 5 (round 4): kfn_conv_desc starts with `struct_size` (the struct had grown at its end -- weights_path -- without a
 * version bump); kfn_comm_rank asks RCCL; kfn_kalman_scan_ex no longer allocates.  A host checks
 * kfn_abi_version() == KFN_ABI_VERSION once after loading the library.
 * 6 (round 5): split-K Winograd entry points, kfn_winograd_lds_bytes.  7 (round 5): kfn_decode_png_rgb8.
 * 8 (round 6): KFN_WINO_FORM_S2_F42, kfn_apply_transform / kfn_pixel_map / kfn_bilinear_sampler.
 * 9 (round 6): kfn_conv_desc.x_layout / y_layout (KFN_LAYOUT_C16).  10 (round 6): kfn_kalman_arith_probe. */
#define KFN_ABI_VERSION 10

const char* kfn_last_error(void);
int kfn_abi_version(void);
/* Device facts used by the host-side tile heuristics.  arch receives e.g. "gfx950". */
int kfn_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len);

/* ---- host side of the image stream ---------------------------------------------------------------------------------
 * tf.image.decode_png(channels=3) of the reference's queue runners (KFNet/train.py:195-239, the decode at :213-217):
 * n PNG files -> dst [n][H][W][3] uint8 RGB, decoded by `threads` native threads (<= 0: one per file, at most the
 * hardware's) straight into the caller's staging buffer (page-locked or not).  Host code only -- no device access, no GIL.
 * Non-interlaced files of bit depth <= 8, colour types gray / RGB / palette / gray+alpha / RGBA (alpha dropped, gray
 * replicated: what decode_png(channels=3) and PIL's convert('RGB') return).  status [n] (may be NULL) receives one of the
 * KFN_PNG_* codes per file: an interlaced or 16-bit file (or one without the PNG signature) is KFN_PNG_UNSUPPORTED -- its frame in dst is left untouched and
 * the call still succeeds (kfnet_amd.pipeline decodes those with PIL); a missing, corrupt or wrong-sized file is
 * KFN_PNG_ERROR and the call returns KFN_ERR_ARG with kfn_last_error() naming the first such file. */
#define KFN_PNG_OK 0
#define KFN_PNG_UNSUPPORTED 1
#define KFN_PNG_ERROR 2
int kfn_decode_png_rgb8(const char* const* paths, int n, int H, int W, unsigned char* dst, int* status, int threads);

/* ---- plumbing for hosts that do not bring their own allocator/streams (PyTorch does) -- */
int kfn_malloc(void** dptr, size_t bytes);
int kfn_free(void* dptr);
int kfn_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int kfn_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int kfn_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int kfn_memset(void* dst, int value, size_t bytes, void* stream);
int kfn_stream_create(void** stream);
int kfn_stream_destroy(void* stream);
int kfn_stream_sync(void* stream);
int kfn_event_create(void** event);
int kfn_event_destroy(void* event);
int kfn_event_record(void* event, void* stream);
int kfn_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */

/* ---- convolution: replaces tf.layers.conv2d / tf.layers.conv2d_transpose -------------
 * Network.conv   cnn_wrapper/network.py:116-135  (SCoordNet.py:21-32, OFlowNet.py:19-41,
 *                KFNet/KFNet.py:318-338 call tf.layers.conv2d directly)
 * Network.deconv cnn_wrapper/network.py:418-437  (OFlowNet.py:26,31,36)
 * tf.layers.dense (OFlowNet.py:50-55) is the 1x1 case on a [P,1,1,C] tensor.
 *
 * Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, fp32 accumulate):
 *   M = N*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin (requires Cin % 16 == 0).
 * Weights are PRE-PACKED by the host into w_packed[cout_pad][kh*kw*Cin] (K contiguous;
 * row co holds w[kh][kw][ci][co] in (kh,kw,ci) order; for transposed convs the TF
 * [kh,kw,Cout,Cin] kernel is packed in the same (kh,kw,ci) order); cout_pad is Cout
 * rounded up to a multiple of 32 with zero rows.  Padding is TF 'SAME'.
 */
typedef struct kfn_conv_desc {
  int32_t struct_size;  /* = sizeof(kfn_conv_desc) AS THE CALLER COMPILED IT (use KFN_CONV_DESC_INIT).  The struct only
                         * ever grows at its end: the library copies struct_size bytes and reads every field beyond them
                         * as 0 (= AUTO / fp32), so a host built against an older header keeps working and the library
                         * never reads past the caller's object.  Smaller than the first-round struct (through `config`),
                         * not a multiple of 4, or larger than the library's own struct: KFN_ERR_ARG. */
  int32_t N, H, W, Cin; /* logical input shape */
  int32_t ldx;          /* input pixel stride (floats), >= Cin, multiple of 4 */
  int32_t Cout;         /* logical output channels */
  int32_t cout_pad;     /* rows of w_packed (multiple of 32) */
  int32_t ldy;          /* output pixel stride (floats), >= Cout */
  int32_t kh, kw;       /* kernel size (kh*kw <= 32) */
  int32_t stride;       /* 1 or 2 */
  int32_t transposed;   /* 0 = conv2d SAME, 1 = conv2d_transpose SAME (stride 2) */
  int32_t relu;         /* fuse tf.nn.relu */
  int32_t epilogue;     /* KFN_EPI_* applied after bias(+relu) */
  int32_t config;       /* 0 = auto tile choice, else KFN_CFG_* */
  int32_t operand_dtype; /* KFN_OPERAND_F32 (exact fp32 MFMA) or KFN_OPERAND_F16: operands rounded
                          * to fp16 while staged, fp32 accumulate on v_mfma_f32_32x32x16_f16
                          * (BASELINE config 5); then w_packed holds IEEE halfs and Cin % 32 == 0.
                          * Activations and outputs stay fp32 in memory unless x_dtype / y_dtype say
                          * otherwise. */
  int32_t wino_order;    /* Winograd kernels only: workgroup order, KFN_WINO_ORDER_* (0 = the kernel's default) */
  int32_t wino_form;     /* kfn_conv2d_winograd_fused / _f43 / _s2: KFN_WINO_FORM_* (0 = auto) */
  int32_t x_dtype;       /* KFN_ACT_F32 / KFN_ACT_F16: element type of the input activations in memory */
  int32_t y_dtype;       /* ... of the output activations.  KFN_ACT_F16 needs operand_dtype == KFN_OPERAND_F16
                          * (BASELINE config 5: fp16 activations end to end); ldx / ldy count ELEMENTS. */
  int32_t k_step;        /* 0 = auto; 16 / 32 = LDS k-step in 4-byte words (fp16 operands: 32 / 64 channels per
                          * stage).  32 exists for the fp16-activation kernels only. */
  int32_t weights_path;  /* fp16 activations in AND out only: KFN_WEIGHTS_AUTO / _VIA_REGISTERS (global -> registers ->
                          * ds_write) / _LDS_DMA (global -> LDS directly, `buffer_load ... lds`, three weight buffers) /
                          * KFN_OPERANDS_LDS_DMA (the activation tile too) */
  int32_t x_layout;      /* KFN_LAYOUT_NHWC (0) / KFN_LAYOUT_C16: memory layout of the input activations (ABI 9) */
  int32_t y_layout;      /* ... of the output.  Only kfn_conv2d_winograd_f43 (eight-wave form) and kfn_conv2d_winograd_s2
                          * (F(4,2) form) take KFN_LAYOUT_C16; every other entry point refuses a non-zero layout
                          * (KFN_ERR_UNSUPPORTED) instead of reading the buffer as NHWC. */
} kfn_conv_desc;
/* kfn_conv_desc d = KFN_CONV_DESC_INIT;  -- zero everything, set struct_size */
#define KFN_CONV_DESC_INIT {(int32_t)sizeof(kfn_conv_desc)}

#define KFN_WEIGHTS_AUTO 0
#define KFN_WEIGHTS_VIA_REGISTERS 1
#define KFN_WEIGHTS_LDS_DMA 2
#define KFN_OPERANDS_LDS_DMA 3    /* activations AND weights global -> LDS directly, three buffers each */

#define KFN_ACT_F32 0
#define KFN_ACT_F16 1

/* Activation layouts.  NHWC: element (n, h, w, c) at ((n*H + h)*W + w)*ld + c -- the layout of the reference's tensors and the
 * default everywhere.  C16 ("channel-blocked"): per image [C/16][H][W][16], element at n*H*W*C + (((c/16)*H + h)*W + w)*16
 * + c%16 -- needs C % 16 == 0 and a dense tensor (ld == C).  The image stride is the same, so batch windows of a tensor are
 * layout-agnostic.  The Winograd kernels read 16 input channels of a patch per step: in C16 a 6-pixel patch row is 384
 * contiguous bytes instead of six 64-byte pieces of six different lines (measured: the F(4x4,3x3) kernel -2 ... -3 %). */
#define KFN_LAYOUT_NHWC 0
#define KFN_LAYOUT_C16 1

/* Order in which the Winograd kernels' workgroups walk (tile block, channel group): with the tile blocks
 * fastest every resident workgroup of an XCD streams the same weight slice and the input crosses the
 * fabric once per channel group; with the channel groups fastest the input crosses once and every group's
 * weights are live at once.  Run time is the same either way (profiles/r03_wino_order_ab.log); the field
 * exists so that this A/B stays reproducible without hidden state. */
#define KFN_WINO_ORDER_AUTO 0
#define KFN_WINO_ORDER_M_FAST 1
#define KFN_WINO_ORDER_N_FAST 2
#define KFN_WINO_ORDER_GROUPS(n) (16 + (n)) /* kfn_conv2d_winograd_f43 / _s2: n channel groups (of 64 / 128) of a tile block adjacent */
#define KFN_WINO_FORM_AUTO 0
#define KFN_WINO_FORM_ONE_WAVE 1 /* force wino2_kernel (one wave per 32 output channels) where wino3_kernel / wino3_pair_kernel would run */
#define KFN_WINO_FORM_F43_FOUR_WAVE 2  /* kfn_conv2d_winograd_f43: wino4_kernel (four waves, 32x32x2 MFMA tiles) */
#define KFN_WINO_FORM_F43_EIGHT_WAVE 3 /* kfn_conv2d_winograd_f43: wino4b_kernel (eight waves, 16x16x4 MFMA tiles) */
#define KFN_WINO_FORM_S2_EIGHT_WAVE 4  /* kfn_conv2d_winograd_s2: wino_s2b_kernel (eight waves, 16x16x4 MFMA tiles; fp32 operands) */
#define KFN_WINO_FORM_S2_F42 5         /* kfn_conv2d_winograd_s2: wino_s2c_kernel, polyphase + F(4,2) on 4x4 output tiles (81 instead of 100
                                        * products per 16 outputs; fp32, H and W multiples of 8; weights: pack_winograd_s2_kernel_c) */

#define KFN_OPERAND_F32 0
#define KFN_OPERAND_F16 1
/* fp32-class accuracy on the fp16 MFMA: operands split into hi + lo halfs while staged,
 * hi*hi + hi*lo + lo*hi accumulated in fp32 (forward convs, Cin % 32 == 0).  w_packed =
 * [hi | lo] half matrices of 1024*w (each [cout_pad][K]); the kernel scales the sum by 2^-10. */
#define KFN_OPERAND_F16X3 2

#define KFN_EPI_NONE 0
#define KFN_EPI_L2NORM 1   /* tf.nn.l2_normalize(axis=-1), KFNet/KFNet.py:340; needs Cout == 32 */
#define KFN_EPI_EXP_CH3 2  /* SCoordNet.GetOutput: uncertainty = exp(ch 3), SCoordNet.py:39-44 */
#define KFN_EPI_EXP_1E2 3  /* OFlowNet.GetOutput: exp(.) * 1e-2, OFlowNet.py:56 */

#define KFN_CFG_AUTO 0
#define KFN_CFG_160x128 1  /* 4 waves, wave tile 160x32 */
#define KFN_CFG_128x128 2  /* 4 waves, wave tile 64x64 */
#define KFN_CFG_128x64 3   /* 4 waves, wave tile 64x32 */
#define KFN_CFG_128x32 4   /* 4 waves, wave tile 32x32 */
#define KFN_CFG_64x64 5    /* 4 waves, wave tile 32x32 */
#define KFN_CFG_256x32 6   /* 4 waves, wave tile 64x32 (Cout <= 32 layers) */
#define KFN_CFG_192x64 7   /* 4 waves, wave tile 96x32 (Cout == 64 layers) */
#define KFN_CFG_160x256 8  /* 8 waves, wave tile 160x32 (never chosen automatically) */
#define KFN_CFG_128x256 9  /* 4 waves, wave tile 64x128: the Winograd GEMMs' tile (fewest loads per MFMA) */
#define KFN_CFG_256x16 10  /* 16-column tiles on v_mfma_f32_16x16x4_f32 for the 16-channel layers */
#define KFN_CFG_128x16 11  /* (fp32 operands, no fused head epilogue)                               */
#define KFN_CFG_256x64 12  /* 4 waves side by side in M, wave tile 64x64: the fp16-activation kernels on 64-channel layers */
#define KFN_CFG_256x256 13    /* fp16 activations only: 4 waves, wave tile 128x128 (16 accumulators = 256 registers, one wave
                               * per SIMD): half the LDS fragment reads and half the operand staging per MFMA of 128x256 */
#define KFN_CFG_256x256_W8 14 /* fp16 activations only: 8 waves (2 x 4), wave tile 128x64, two waves per SIMD */
#define KFN_CFG_512x64 15     /* fp16 activations only: 4 waves side by side in M, wave tile 128x64 (64-channel layers) */

int kfn_conv2d_nhwc(const kfn_conv_desc* desc, const float* x, const float* w_packed,
                    const float* bias /* [Cout] or NULL */, float* y, void* stream);
/* 3x3 stride-1 SAME convolution 64 -> 64 channels on fp16 activations (x_dtype = y_dtype = KFN_ACT_F16, operand_dtype =
 * KFN_OPERAND_F16; SCoordNet's conv1b in BASELINE config 5, cnn_wrapper/SCoordNet.py:20): the weights stay in registers for
 * a workgroup's life, the workgroup walks down a strip of 128 / 192 pixels and reads every input row from the LDS once for
 * the three output rows it feeds (csrc/kfn_conv64.hip).  Same arithmetic as kfn_conv2d_nhwc on that descriptor (fp16
 * products, fp32 accumulation, bias, ReLU, one RNE rounding), another summation order.  w_packed = [2][36][64][8] halfs
 * (kfnet_amd.graph.pack_conv64_rows_kernel); x, y, w_packed, bias 16-byte aligned; ldx, ldy multiples of 8 elements.
 * kfn_conv3x3_c64_f16_supported(desc) = 1 when the launch takes the layer (KFN_ERR_UNSUPPORTED otherwise). */
int kfn_conv3x3_c64_f16_supported(const kfn_conv_desc* desc);
int kfn_conv3x3_c64_f16(const kfn_conv_desc* desc, const void* x, const void* w_packed, const float* bias, void* y,
                        void* stream);
/* Output spatial size for a descriptor (TF SAME rule); host-side shape inference. */
int kfn_conv2d_out_shape(const kfn_conv_desc* desc, int* Ho, int* Wo);
/* Which kernel instantiation kfn_conv2d_nhwc will launch for `desc`: tile config
 * (KFN_CFG_*), k-step (16|32) and workgroup count.  Used by bench.py to attribute
 * measured launch times to kernel names. */
int kfn_conv2d_plan(const kfn_conv_desc* desc, int* config, int* bk, int* tiles);

/* Which entry points the default graph (kfnet_amd.KFNet / KFNetEngine) launches, and which it does not
 *   ON the default route: kfn_first_conv_u8[_ex], kfn_conv2d_nhwc, kfn_conv3x3_c64_f16 (config 5), kfn_conv2d_winograd_fused, kfn_conv2d_winograd_f43,
 *     kfn_conv2d_winograd_s2, kfn_pad_nhwc, kfn_oflow_head[_f16], kfn_oflow_tail2[_f16], kfn_kalman_scan[_ex], kfn_eval_metrics,
 *     kfn_send_state / kfn_recv_state (multi-GPU), kfn_copy_channels (concat fallback).
 *   LEGACY -- earlier forms of the same operators, superseded on the default route, kept as tested stand-alone
 *     operators (and reachable through the Graph switches named in DESIGN.md): kfn_conv2d_winograd (+
 *     kfn_winograd_workspace_bytes / kfn_winograd_plan: the two-kernel Winograd form), kfn_cost_volume (materialising),
 *     kfn_cost_volume_conv (volume generated in conv0's loader), kfn_cost_volume_gather, kfn_flow_softargmax,
 *     kfn_flow_head, kfn_oflow_tail.  A maintainer wiring the reference to this library needs none of them.
 *   Reference API beside eval.py's path: kfn_kalman_fuse, kfn_kalman_fuse2 (KFNet.BuildKFCoord / GetKFCoord2 alone). */

/* [LEGACY: two-kernel form]  Winograd F(2x2,3x3) variant for 3x3 stride-1 SAME convolutions (same arguments and
 * result as kfn_conv2d_nhwc up to fp32 round-off, 2.25x fewer MFMA FLOPs): 16 GEMMs on the
 * MFMA kernel with the B^T d B input transform evaluated in its loader, then A^T M A +
 * bias + ReLU.  u_packed = [16][cout_pad][Cin], group g = 4*xi+nu holding (G g G^T)[xi][nu]
 * transposed to [co][ci]; workspace >= kfn_winograd_workspace_bytes(desc).
 * phases: 3 = whole convolution; 1 = only the 16 GEMMs, 2 = only the output transform
 * (lets a profiler time the two kernels separately). */
int kfn_winograd_workspace_bytes(const kfn_conv_desc* desc, size_t* bytes);
int kfn_winograd_plan(const kfn_conv_desc* desc, int* config, int* bk, int* tiles); /* cf. kfn_conv2d_plan */
int kfn_conv2d_winograd(const kfn_conv_desc* desc, const float* x, const float* u_packed,
                        const float* bias, float* y, float* workspace, int phases, void* stream);

/* Single-kernel variant of the above (csrc/kfn_wino2.hip): one wavefront = one workgroup owns a
 * block of 8x4 Winograd tiles x 32 output channels x all 16 (xi,nu) positions (16 accumulators
 * of 32x32 in registers), evaluates B^T d B on raw input patches staged once through LDS and
 * A^T M A in registers -- no workspace, no second kernel, every input pixel fetched once per
 * workgroup.  Same result as kfn_conv2d_winograd up to fp32 summation order.
 * u2_packed = [Cin/8][16][cout_pad][8]: the (G g G^T)[xi][nu] of kfn_conv2d_winograd, re-packed so
 * that the 32-channel fragment of one (8-channel k-chunk, position) is one contiguous 1 KiB read:
 *   u2[((ci/8)*16 + 4*xi+nu)*cout_pad + co][ci%8].
 * Needs Cin % 16 == 0, (H+1)/2 >= 4, cout_pad % 32 == 0, no fused head epilogue
 * (kfn_winograd_fused_supported() == 1); KFN_ERR_UNSUPPORTED otherwise.  Layers with Cout >= 128 and
 * Cin % 32 == 0 run in the four-wave form (csrc/kfn_wino3.hip: one input transform per 128 output channels,
 * shared through LDS); layers with 33 .. 64 output channels and Cin % 16 == 0 in its two-wave form (one transform
 * and one read of the input for all their channels; two images of the input below 1 GiB).  operand_dtype KFN_OPERAND_F16 (BASELINE config 5; four-wave form only, Cin % 64 == 0): u2_packed holds
 * IEEE halfs in the same layout, the input transform runs in fp32 on the fp32 activations and is rounded to fp16
 * when it is shared, the products are fp16 MFMAs with fp32 accumulation. */
int kfn_winograd_fused_supported(const kfn_conv_desc* desc);
int kfn_conv2d_winograd_fused(const kfn_conv_desc* desc, const float* x, const void* u2_packed,
                              const float* bias, float* y, void* stream);

/* 3x3 STRIDE-2 'same' convolution of an image with even H and W (tf.layers.conv2d(3, strides=2, 'same'),
 * cnn_wrapper/network.py:116-135: SCoordNet conv2a / conv3a / conv4a, cnn_wrapper/SCoordNet.py:12-27) by
 * polyphase decomposition + F(2,2) minimal filtering: 25 instead of 36 multiplies per 2x2 outputs, one launch,
 * no workspace (csrc/kfn_wino_s2.hip).  u2_packed = the 16 pre-transformed, pre-signed weight fragments
 * [Cin/8][16][cout_pad][8] (fragments 0-8: G g00 G^T of the taps w[2a][2b]; 9-11: G (w[0][1], w[2][1]);
 * 12-14: G (w[1][0], w[1][2]); 15: w[1][1]; the fragments of Winograd index 2 negated) -- kfnet_amd.graph.
 * pack_winograd_s2_kernel.  kfn_conv_desc.wino_form = KFN_WINO_FORM_S2_EIGHT_WAVE launches wino_s2b_kernel instead (two waves
 * per SIMD on 16x16x4 MFMA tiles, fp32 operands only, 1.5-2.5 % faster: what the default graph does); its weights are packed
 * per PAIR of fragments: u2b[((ci/8)*8 + f/2)*cout_pad + co][4*((ci%8)/2) + 2*(f%2) + ci%2]
 * (kfnet_amd.graph.pack_winograd_s2_kernel_b).  Needs Cin % 16 == 0, H and W even, H >= 14, cout_pad % 32 == 0, no fused head
 * epilogue (kfn_winograd_s2_supported() == 1); KFN_ERR_UNSUPPORTED otherwise.  operand_dtype KFN_OPERAND_F16
 * (BASELINE config 5): u2_packed holds IEEE halfs in the same layout, the input transform runs in fp32 and is
 * rounded to fp16 when it is shared, fp16 MFMAs with fp32 accumulation. */
int kfn_winograd_s2_supported(const kfn_conv_desc* desc);
int kfn_conv2d_winograd_s2(const kfn_conv_desc* desc, const float* x, const void* u2_packed, const float* bias,
                           float* y, void* stream);
/* Split-K form of the eight-wave stride-2 kernel (fp32 operands, u2b_packed = the pair packing; Cout % 4 == 0, ldy % 4 == 0,
 * y 16-byte aligned) for launches that leave the chip partly idle: semantics, workspace layout ([k_split][N*Ho*Wo][Cout]
 * floats) and determinism exactly as kfn_conv2d_winograd_f43_splitk below; the planes are reduced by the same second kernel. */
int kfn_winograd_s2_splitk_workspace_bytes(const kfn_conv_desc* desc, int k_split, size_t* bytes);
int kfn_conv2d_winograd_s2_splitk(const kfn_conv_desc* desc, const float* x, const void* u2b_packed, const float* bias,
                                  float* y, float* workspace, int k_split, void* stream);

/* Winograd F(4x4,3x3) for the same 3x3 stride-1 'same' convolutions (csrc/kfn_wino4.hip, round 4): 36 products per
 * 4x4 outputs -- 2.25 multiplies per output instead of 4 (F(2x2,3x3)) or 9 (direct) -- interpolation points
 * {0, +-1, +-2}, fp32 throughout, one launch, no workspace.  Meant for the layers with Cin >= 512 (SCoordNet conv3b /
 * conv4b / conv5, cnn_wrapper/SCoordNet.py:26-30), where the K loop amortises the larger transforms; the result differs
 * from kfn_conv2d_nhwc by ~3x the F(2x2,3x3) round-off (1.4e-6 on the network's coordinates, tools/experiments/
 * f43_error_budget.py).  u4_packed = [Cin/8][36][cout_pad][8]: U[6 xi + nu] = (G g G^T)[xi][nu] in fp64, rounded once,
 * u4[((ci/8)*36 + 6*xi+nu)*cout_pad + co][ci%8] (kfnet_amd.graph.pack_winograd_f43_kernel).
 * Two kernels behind the entry point, chosen by kfn_conv_desc.wino_form, and the weight layout goes with the choice:
 *   KFN_WINO_FORM_AUTO / KFN_WINO_FORM_F43_FOUR_WAVE: wino4_kernel, four waves on 32x32x2 MFMA tiles, the layout above;
 *   KFN_WINO_FORM_F43_EIGHT_WAVE: wino4b_kernel, two waves per SIMD on 16x16x4 tiles (3-13 % faster: what the default graph
 *     launches), u4_packed = [Cin/8][18 position pairs][cout_pad][4 k][2 positions][2 k-steps]:
 *     u4b[((ci/8)*18 + p/2)*cout_pad + co][4*((ci%8)/2) + 2*(p%2) + ci%2], p = 6*xi+nu
 *     (kfnet_amd.graph.pack_winograd_f43_kernel_b) -- same U, same size, a lane's operands of two positions contiguous.
 * Needs Cin % 16 == 0, H >= 29, Cout % 4 == 0, ldy % 4 == 0, ldx % 2 == 0, y 16-byte aligned, two images of the input
 * below 1 GiB, fp32 operands and activations, no fused head epilogue (kfn_winograd_f43_supported() == 1);
 * KFN_ERR_UNSUPPORTED otherwise. */
int kfn_winograd_f43_supported(const kfn_conv_desc* desc);
/* Dynamic LDS (bytes per workgroup) of the kernel the matching Winograd entry point launches for `desc`: stride 2 ->
 * kfn_conv2d_winograd_s2 (wino_form AUTO / KFN_WINO_FORM_S2_EIGHT_WAVE), stride 1 with wino_form KFN_WINO_FORM_F43_* ->
 * kfn_conv2d_winograd_f43, any other stride-1 descriptor -> the form kfn_conv2d_winograd_fused routes it to.  A host
 * compares it with kfn_device_info()'s lds_bytes_per_cu and routes around kernels the device cannot hold (147-157 KB for
 * the F(4x4,3x3) and four-wave F(2x2,3x3) forms) instead of failing in the first launch.  No device access.  (ABI 6) */
int kfn_winograd_lds_bytes(const kfn_conv_desc* desc, int* bytes);
/* Split-K form of the eight-wave kernel for launches that would leave the chip idle (BASELINE configs[1], one 480x640 frame:
 * SCoordNet conv4b / conv5 / conv6 at batch 1 are 160 / 80 / 40 workgroups on 256 CUs, cnn_wrapper/SCoordNet.py:27-30): the
 * input channels are cut into k_split runs of ceil(Cin / 16 / k_split) super-steps, k_split copies of the tile grid write raw
 * partial sums into the planes of `workspace` ([k_split][N*H*W][Cout] floats, kfn_winograd_f43_splitk_workspace_bytes), a
 * second kernel adds the planes in the fixed order 0 .. k_split-1, then bias and ReLU, into y -- two stream-ordered launches,
 * no atomics, bit-stable from run to run.  u4b_packed = the eight-wave packing (kfnet_amd.graph.pack_winograd_f43_kernel_b);
 * 1 <= k_split <= Cin / 16 with no empty run; k_split == 1 is kfn_conv2d_winograd_f43's eight-wave launch (workspace unused).
 * The result differs from the unsplit launch by the summation order over the input channels only. */
int kfn_winograd_f43_splitk_workspace_bytes(const kfn_conv_desc* desc, int k_split, size_t* bytes);
int kfn_conv2d_winograd_f43_splitk(const kfn_conv_desc* desc, const float* x, const float* u4b_packed, const float* bias,
                                   float* y, float* workspace, int k_split, void* stream);
int kfn_conv2d_winograd_f43(const kfn_conv_desc* desc, const float* x, const float* u4_packed, const float* bias,
                            float* y, void* stream);

/* ---- first layers: uint8 image -> (x-128)*0.00625 -> 3x3 conv, Cin = 3 --------------
 * Replaces SCoordNet.preprocess + conv1a (SCoordNet.py:20-21,34-37) and the feature
 * tower's preprocess + feat1 (KFNet/KFNet.py:317-320) in ONE pass over the image.
 * w1 [27][C1], w2 [27][C2] = the TF HWIO kernels flattened (kh,kw,ci major, co minor).
 * C1, C2 multiples of 16; C2 may be 0 (then w2/b2/y2 are ignored).  ReLU is fused. */
int kfn_first_conv_u8(const uint8_t* img, int N, int H, int W,
                      const float* w1, const float* b1, float* y1, int C1,
                      const float* w2, const float* b2, float* y2, int C2, void* stream);
/* The same with the FIRST head's output element type chosen (KFN_ACT_F32 | KFN_ACT_F16): BASELINE config 5 keeps
 * SCoordNet's activations in fp16 from conv1a on (C1 == 64); the second head (feat1) stays fp32. */
int kfn_first_conv_u8_ex(const uint8_t* img, int N, int H, int W,
                         const float* w1, const float* b1, void* y1, int C1, int y1_dtype,
                         const float* w2, const float* b2, float* y2, int C2, void* stream);

/* ---- [LEGACY: materialising]  KFNet.BuildCoordVolume + reshape (KFNet/KFNet.py:343-359, :372) ----
 * vol[n, y, x, i, j, c] = f2[n,y,x,c] - f1[n, y+i-w/2, x+j-w/2, c]  (0 outside).
 * f1, f2: [N,H,W,C] (C % 4 == 0); vol: [N*H*W, window, window, C]. */
int kfn_cost_volume(const float* f1, const float* f2, float* vol, int N, int H, int W, int C,
                    int window, void* stream);

/* [LEGACY: Graph.factor_cost_volume = False]  BuildCoordVolume fused into OFlowNet's first conv (cnn_wrapper/OFlowNet.py:19, 3x3 SAME on
 * the 8x8 window grid): y[p, i, j, :] = act(conv0(vol)[p, i, j, :]) with vol generated inside
 * the MFMA kernel's loader (window fixed at 8, kernel 3x3); the 39 MB/frame volume never
 * exists in HBM.  f1, f2 [N,H,W,C] (C % 16 == 0); w_packed [cout_pad][9*C] as for
 * kfn_conv2d_nhwc; y [N*H*W, 8, 8, Cout] with pixel stride ldy. */
int kfn_cost_volume_conv(const float* f1, const float* f2, const float* w_packed, const float* bias,
                         float* y, int N, int H, int W, int C, int Cout, int cout_pad, int ldy,
                         int relu, int config, void* stream);

/* ---- factored cost volume + conv0 ---------------------------------------------------------
 * conv0 (cnn_wrapper/OFlowNet.py:19) is linear and V[p,cell] = f2[p] - f1[p+cell-4]
 * (KFNet/KFNet.py:343-359), hence conv0(V)[p,ci,cj] = b + S_k f2[p] - G_k(p+(ci-4,cj-4)) with
 * k = 3*rowclass(ci) + colclass(cj) (which taps stay inside the 8x8 window), S_k the sum of
 * those taps and G_k the 3x3 SAME conv of the zero-extended f1 with them.  T = b + S f2
 * [N,H,W,9*C] and Gp = G on the map extended by 2 [N,H+4,W+4,9*C] come from kfn_conv2d_nhwc
 * (kfn_pad_nhwc pads f1 by 2); this launch gathers, subtracts and applies the ReLU into the
 * [(N*H*W),8,8,C] tensor conv0 used to produce (pixel stride ldy). */
int kfn_pad_nhwc(const float* x, float* y, int N, int H, int W, int C, int pad, void* stream);
/* [LEGACY: Graph.fuse_oflow_window = False -- kfn_oflow_head / kfn_oflow_tail2 evaluate T - G where they need it] */
int kfn_cost_volume_gather(const float* T, const float* Gp, float* y, int N, int H, int W, int C,
                           int ldy, int relu, void* stream);

/* ---- [LEGACY: debug_prob path]  softmax over the window cells + soft-argmax flow ---------
 * OFlowNet.GetOutput softmax (OFlowNet.py:45-47) + KFNet.BuildOFlowNet flow
 * (KFNet/KFNet.py:381-385): flow[p] = sum_k softmax(logits[p])_k * (j-w/2, i-w/2).
 * logits [P, window*window] (row-major i,j); flow_xy [P,2]; prob [P,window^2] or NULL. */
int kfn_flow_softargmax(const float* logits, float* flow_xy, float* prob, int P, int window,
                        void* stream);

/* ---- [LEGACY: Graph.fuse_oflow_tail = False]  fused flow head: 'prediction' conv + softmax + soft-argmax ----
 * cnn_wrapper/OFlowNet.py:41 (3x3 conv C->1, SAME, no ReLU on the 8x8 window grid) +
 * OFlowNet.py:45-47 + KFNet/KFNet.py:381-385 in one kernel; the 64 logits stay on chip.
 * x [P,8,8,C] (C % 4 == 0, C <= 32); w [3,3,C] = the TF kernel [3,3,C,1] flattened;
 * bias [1] or NULL; flow_xy [P,2]; opt_logits [P,64] or NULL (debug). */
int kfn_flow_head(const float* x, const float* w, const float* bias, float* flow_xy,
                  float* opt_logits, int P, int C, void* stream);

/* [LEGACY: Graph.fuse_oflow_window = False]  OFlowNet's tail for every window in ONE launch: conv6 (3x3, 48 -> 16, ReLU; cnn_wrapper/OFlowNet.py:36-40) on
 * x = concat0 [P,8,8,48], the 'prediction' conv (3x3, 16 -> 1, linear; OFlowNet.py:41), the softmax over the 64
 * cells (OFlowNet.py:45-47) and the soft-argmax flow (KFNet/KFNet.py:381-385) -> flow_xy [P,2] (and the 64 logits
 * per window into opt_logits when given).  One wave per window, the patch resident in LDS, conv6's weights in
 * registers (csrc/kfn_oflow_tail.hip).  w6_packed = [108][64] per-lane fragments (kfnet_amd.graph.
 * pack_oflow_tail_kernel), wp = [3][3][16].  Only c_in = 48, c_mid = 16 (KFN_ERR_UNSUPPORTED otherwise). */
int kfn_oflow_tail(const float* x, const float* w6_packed, const float* b6, const float* wp, const float* bp,
                   float* flow_xy, float* opt_logits, int P, int c_in, int c_mid, void* stream);

/* ---- OFlowNet's two window-grid ends, window-resident (csrc/kfn_oflow_fused.hip) -------------------------
 * Both take the per-pixel maps of the factored cost volume (see kfn_cost_volume_gather): T [N,H,W,9*32] and
 * Gp [N,H+4,W+4,9*32], and evaluate conv0's cells (cnn_wrapper/OFlowNet.py:19 on the volume of
 * KFNet/KFNet.py:343-359) where they need them -- conv0's [N*H*W,8,8,32] output never exists in memory.
 *   kfn_oflow_head : conv0 -> conv1a (3x3 stride 2, 32 -> 32, ReLU; OFlowNet.py:20) -> y [N*H*W,4,4,32].
 *                    w1_packed [144][64] per-lane fragments (kfnet_amd.graph.pack_oflow_head_kernel).
 *   kfn_oflow_tail2: upconv0 (conv2d_transpose 3x3 stride 2, 32 -> 16, ReLU; OFlowNet.py:37) of x5 = conv5's
 *                    output [N*H*W,4,4,32], concat0 = [upconv0 | conv0], conv6, 'prediction', softmax, soft-argmax
 *                    (as kfn_oflow_tail) -> flow_xy [N*H*W,2].  wu_packed [72][64] (pack_oflow_upconv_kernel),
 *                    w6_packed / wp as for kfn_oflow_tail.
 * relu0 = conv0's activation flag.  One wave per window; all weights in registers; exact fp32 MFMAs. */
int kfn_oflow_head(const float* T, const float* Gp, int N, int H, int W, int relu0, const float* w1_packed,
                   const float* b1, float* y, void* stream);
/* kfn_oflow_head for BASELINE config 5: conv1a multiplies on v_mfma_f32_16x16x16_f16 (operands rounded to halfs where the
 * MFMA reads them / by the packer; conv0's T - G, the accumulation, bias and ReLU fp32; y fp32).  w1_packed_f16 =
 * [36][64][4] halfs (kfnet_amd.graph.pack_oflow_head_kernel_f16). */
int kfn_oflow_head_f16(const float* T, const float* Gp, int N, int H, int W, int relu0, const void* w1_packed_f16,
                       const float* b1, float* y, void* stream);
int kfn_oflow_tail2(const float* T, const float* Gp, int N, int H, int W, int relu0, const float* x5,
                    const float* wu_packed, const float* bu, const float* w6_packed, const float* b6,
                    const float* wp, const float* bp, float* flow_xy, float* opt_logits, void* stream);
/* kfn_oflow_tail2 for BASELINE config 5 ("fp16 convs"): upconv0 and conv6 multiply on v_mfma_f32_16x16x16_f16 -- their
 * operands are rounded to IEEE halfs (activations where the MFMA reads them, to nearest even; the weights by the packer), the
 * accumulation, conv0's T - G, the ReLUs, 'prediction', softmax and soft-argmax stay fp32.  wu_packed_f16 = [18][64][4]
 * halfs (kfnet_amd.graph.pack_oflow_upconv_kernel_f16), w6_packed_f16 = [27][64][4] halfs (pack_oflow_tail_kernel_f16);
 * every other argument as for kfn_oflow_tail2. */
int kfn_oflow_tail2_f16(const float* T, const float* Gp, int N, int H, int W, int relu0, const float* x5,
                        const void* wu_packed_f16, const float* bu, const void* w6_packed_f16, const float* b6,
                        const float* wp, const float* bp, float* flow_xy, float* opt_logits, void* stream);

/* ---- the recurrent part: warp + Kalman predict/update + NIS + transform/emit ----------
 * One launch scans T frames of S independent sequences (one workgroup per sequence,
 * state resident in LDS when it fits).  Per frame and pixel, in this order:
 *   KFNet.BuildOFlowNet tail (KFNet/KFNet.py:386-401): pixel_map = (x,y)+flow;
 *     x^- = bilinear(last_coord), s_l = bilinear(last_unc) (tools/util.py:36-93,
 *     clamped corners AND clamped-corner weights); P^- = max(s_l^2,eps^2)+max(s_t^2,eps^2)
 *   KFNet.BuildKFCoord (KFNet/KFNet.py:148-162): K = P^-/(P^-+R), x = max(1-K,0) x^- + K z
 *   KFNet.GetNIS (KFNet/KFNet.py:164-184)
 *   eval.py:87-126: reset when (t0+t) % reset_period == 0 (state := measurement),
 *     optional NIS gate on the OUTPUT only, record = (T.x, 1/sigma) via ApplyTransform
 *     (KFNet/util.py:12-40).
 * Grids above 10 240 pixels (state > 160 KB) run one launch per frame with the state
 * ping-ponging between `state` and `scratch`, a second copy of the state the CALLER provides
 * (kfn_kalman_scan_scratch_bytes; 0 bytes -- scratch may be NULL -- for every grid that fits
 * the LDS, which includes BASELINE's 60x80 and 68x120); results are the same function of the
 * inputs.
 */
typedef struct kfn_kalman_desc {
  int32_t S, T, H, W;
  int32_t t0;            /* global index of frame 0 of this call (reset phase) */
  int32_t reset_period;  /* eval.py: spec.sequence_length (500); <=0 = never reset */
  float min_uncertainty; /* KFNet.min_uncertainty = 1e-5 */
  float nis_gate;        /* 0 = off; eval.py --NIS uses 7.815 */
  int32_t has_transform; /* 0 = identity */
  float transform[12];   /* first 3 rows of inv(transform.txt), row-major */
} kfn_kalman_desc;

int kfn_kalman_scan_scratch_bytes(const kfn_kalman_desc* desc, size_t* bytes);
int kfn_kalman_scan(const kfn_kalman_desc* desc,
                    const float* flow_xy,     /* [S,T,H*W,2] */
                    const float* sigma_trans, /* [S,T,H*W]   */
                    const float* meas,        /* [S,T,H*W,4] = (z, sigma_z) */
                    float* state,             /* [S,H*W,4] in/out = (x, sigma) raw KF state */
                    float* records,           /* [S,T,H*W,4] = (T.x, 1/sigma) */
                    float* opt_temp,          /* [S,T,H*W,4] = (x^-, sigma^-) or NULL */
                    float* opt_nis,           /* [S,T,H*W,3] or NULL */
                    void* scratch,            /* kfn_kalman_scan_scratch_bytes() bytes, 16-byte aligned, or NULL when 0 */
                    void* stream);

/* The same scan with the extra debug outputs eval.py's log line is computed from:
 *   opt_kf [S,T,H*W,4]  the raw KF estimate (x, sigma) of every frame BEFORE the NIS gate, the reset
 *                        override and ApplyTransform (what KF_loss / KF_accuracy see, KFNet/train.py:256-257);
 *   raw_on_reset != 0   on a reset step opt_temp / opt_nis / opt_kf hold what the GRAPH computes there from
 *                        the incoming state (KFNet/eval.py:78-83 runs before the host override of :94-101);
 *                        with 0 they hold (z, 0, z) as kfn_kalman_scan writes them.
 * State, records and the NIS gate are unaffected.  kfn_kalman_scan == kfn_kalman_scan_ex(.., NULL, 0, ..). */
int kfn_kalman_scan_ex(const kfn_kalman_desc* desc, const float* flow_xy, const float* sigma_trans,
                       const float* meas, float* state, float* records, float* opt_temp, float* opt_nis,
                       float* opt_kf, int raw_on_reset, void* scratch, void* stream);

/* ---- evaluation numbers of eval.py, reduced on the device --------------------------------
 * KFNet.CoordLossWithUncertainty(downsample=True) x3 (KFNet/KFNet.py:192-232 via KFNet/train.py:252-257),
 * the NIS band count (KFNet/eval.py:10-15) and the distance maps of dist_error (eval.py:17-29) for T
 * frames, one workgroup per frame.  Inputs are the scan's buffers [T,H*W,.] (meas, temp, kf_raw, records,
 * nis as written with raw_on_reset = 1) and `labels` [L,H*W,4] = (gt xyz, mask) already nearest-down-sampled
 * (tf.image.resize_nearest_neighbor: source pixel (8y, 8x)); label_pair [T,2] = label rows of the step's
 * frame pair (KFNet/train.py:67-71: the losses see BOTH, the distances the second); reset_flags [T].
 * stats [T,16]: [0..2] masked loss sums (measure, temporal, KF), [3..5] "inaccurate" pixel counts,
 * [6] valid_pixel = sum(mask_a) + sum(mask_b) + 1, [7] NIS values > 0, [8] of those inside (0.0157, 2.706);
 * loss = stats[k]/stats[6], accuracy = (stats[6] - stats[3+k])/stats[6].  dist_maps [T,3,H*W] in cm
 * (0 where masked); the host takes the medians of the positive entries. */
int kfn_eval_metrics(const float* meas, const float* temp, const float* kf_raw, const float* records,
                     const float* nis, const float* labels, const int32_t* label_pair,
                     const uint8_t* reset_flags, const float* transform12 /* 3x4 row-major or NULL */,
                     int T, int HW, float dist_threshold /* 0.05 */, float min_uncertainty /* 1e-5 */,
                     float* stats, float* dist_maps, void* stream);

/* KFNet.BuildKFCoord on its own (KFNet/KFNet.py:148-162) + optional KFNet.GetNIS
 * (:164-184): pred = (x^-, sigma^-), meas = (z, sigma_z), out = (x, sigma), all [P,4];
 * opt_nis [P,3] or NULL.  48 B/pixel of HBM traffic (32 read + 16 written). */
int kfn_kalman_fuse(const float* pred, const float* meas, float* out, float* opt_nis, long P,
                    void* stream);

/* KFNet.GetKFCoord2 (KFNet/KFNet.py:487-502; not on eval.py's path): the same fusion with the posterior variance in
 * the symmetric form (1-K)^2 P^- + K^2 R.  pred / meas / out packed [P,4] = (x, y, z, sigma) as for kfn_kalman_fuse. */
int kfn_kalman_fuse2(const float* pred, const float* meas, float* out, long P, void* stream);

/* Self-check of the arithmetic inside kfn_kalman_scan (ABI 10): the scan computes its two square roots and two quotients per pixel
 * with the refinement steps of the IEEE forms but without their denormal pre-scaling (csrc/kfn_kalman.hip, sqrt_rn_normal /
 * div_rn_normal: the scan is VALU-bound).  out [n,4] = (lean sqrt(a), sqrtf(a), lean a / b, a / b): the pairs must be equal bit
 * for bit for normal-range operands, and for zero / infinity / NaN as IEEE defines them. */
int kfn_kalman_arith_probe(const float* a, const float* b, float* out, long n, void* stream);

/* ---- the graph-level helpers of the reference as stand-alone launches ----------------------------------------
 * On eval.py's path they are fused into the scan (kfn_kalman_scan: warp, fuse, transform in one kernel); these entry
 * points serve Python-level callers (KFNet/eval.py:57-59 calls ApplyTransform itself).  All tensors NHWC fp32 with a
 * pixel stride `ld_*` (floats) so that channel views of packed buffers can be passed.
 *   kfn_apply_transform   KFNet/util.py:12-40: out[p,0:3] = (T [x;1])[0:3], no perspective divide.  `transform` is a DEVICE
 *                         pointer to one 4x4 (per_batch = 0) or B 4x4 (per_batch = 1) row-major matrices.
 *   kfn_pixel_map         KFNet/util.py:42-63: out[b,y,x] = (x, y); normalize != 0: ((x - u) / focal_x, (y - v) / focal_y).
 *   kfn_bilinear_sampler  tools/util.py:3-94: imgs [B,Hs,Ws,C] sampled at coords [B,Ht,Wt,2] = (x, y) -> out [B,Ht,Wt,C];
 *                         corner indices clamped to the image AND weights taken from the clamped corners (so any sample
 *                         with x < 0 or x >= Ws-1 evaluates to 0, x = Ws-1 included), sum in tf.add_n order. */
int kfn_apply_transform(const float* coords, int ld_in, const float* transform, int per_batch, int B, int H, int W,
                        float* out, int ld_out, void* stream);
int kfn_pixel_map(float* out, int ld_out, int B, int H, int W, int normalize, float u, float v, float focal_x,
                  float focal_y, void* stream);
int kfn_bilinear_sampler(const float* imgs, int ld_img, int B, int Hs, int Ws, int C, const float* coords, int ld_coords,
                         int Ht, int Wt, float* out, int ld_out, void* stream);

/* ---- Network.concat fallback (cnn_wrapper/network.py:316-318): strided channel copy -- */
int kfn_copy_channels(const float* src, int ld_src, float* dst, int ld_dst, int P, int C,
                      void* stream);

/* ---- multi-GPU: rank -> rank hand-off of the recurrent state (RCCL point-to-point) ----
 * No reference counterpart (the reference is single-device: KFNet/train.py:375 is its only
 * device placement).  BASELINE config 4 / SURVEY.md §8(e): a T-frame sequence is cut into
 * contiguous chunks, one per rank = GPU; everything but the scan depends on images only, so
 * the one exchange is the [H,W,4] fp32 state (what KFNet/eval.py:103-104 feeds back through
 * SetVariableByName) from the rank that finished chunk r to the rank that owns chunk r+1.
 * ncclSend / ncclRecv on `stream`, i.e. ordered behind/before the kfn_kalman_scan launches
 * on that stream; no host synchronisation.  librccl is bound at run time (dlopen).
 *   kfn_comm_available  local probe: run it on every rank BEFORE the collective kfn_comm_init, so that a rank
 *                       without RCCL is found while the others can still be told;
 *   kfn_comm_unique_id  rank 0 creates the 128-byte id and distributes it out of band
 *                       (torch.distributed / MPI / a file);
 *   kfn_comm_init       collective over all ranks; binds `device` to the calling thread;
 *   kfn_send_state /    count = H*W*4 floats; a send must be matched by the peer's recv of
 *   kfn_recv_state      the same size (kfnet_amd/dist.py pairs them by the reset rule: a
 *                       chunk that starts on a reset frame receives nothing). */
#define KFN_COMM_ID_BYTES 128
typedef struct kfn_comm kfn_comm;
int kfn_comm_available(void);   /* KFN_OK if librccl can be bound in this process (no communicator, no socket, no device) */
int kfn_comm_unique_id(void* id, size_t bytes /* == KFN_COMM_ID_BYTES */);
int kfn_comm_init(kfn_comm** comm, int rank, int nranks, const void* unique_id, int device);
int kfn_comm_destroy(kfn_comm* comm);
int kfn_comm_rank(const kfn_comm* comm, int* rank, int* nranks);
int kfn_send_state(kfn_comm* comm, int peer, const float* state /* [H,W,4] */, int H, int W, void* stream);
int kfn_recv_state(kfn_comm* comm, int peer, float* state /* [H,W,4] */, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KFNET_HIP_H_ */
