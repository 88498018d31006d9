// Shared helpers for the libkfnet_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cstddef>
#include <atomic>
#include <cstdint>
#include "../../include/kfnet_hip.h"

// Cache policy of the wide (16-byte, whole-line) output stores of the convolution kernels: 0 = default, 2 = the
// non-temporal bit (A/B build: NT_STORE=1 tools/mb/build_hot.sh).
#ifndef KFN_NT_STORE_AUX
#define KFN_NT_STORE_AUX 0
#endif

// Wait states behind a 16-byte buffer store whose data registers may be overwritten next (see buffer_store_b128): s_nop N
// = N + 1 states.  (KFN_STORE_PAD=-1 builds the library WITHOUT the pad: the round-4 behaviour, for tools/debug_conv64.py.)
#ifndef KFN_STORE_PAD
#define KFN_STORE_PAD 7
#endif

namespace kfn {

#if defined(__HIPCC__)
// buffer_store_dwordx4 with an SGPR offset.  The store reads its four data VGPRs during the cycles AFTER it has issued.
// hipcc's hazard table pads a following VALU write of those registers only for stores WITHOUT an SGPR offset (it takes
// the SGPR operand to cost the missing cycle), and reuses the dead data registers at once -- e.g. as the next store's
// per-lane offset.  Measured on gfx950 (round 5, conv64_rows_kernel, profiles/r05_conv64_store_hazard.log): with the
// memory system loaded by another stream, about one launch in ten of 2x540x960 sent, for a few 4-lane groups of one wave,
// the NEXT instruction's result (a byte offset) as the first dword of a 16-byte piece.  So every such store goes through
// here: the data (and the per-lane offset) stay live up to an asm statement behind the store, and that statement spends the
// wait states.
template <int AUX, typename V>
__device__ __forceinline__ void buffer_store_b128(V v, __amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
  typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
  static_assert(sizeof(V) == 16, "buffer_store_b128 stores 16 bytes");
  const u32x4_t d = __builtin_bit_cast(u32x4_t, v);
  __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, voffset, soffset, AUX);
#if KFN_STORE_PAD >= 0
#define KFN_STR2(x) #x
#define KFN_STR(x) KFN_STR2(x)
  asm volatile("s_nop " KFN_STR(KFN_STORE_PAD) : : "v"(d), "v"(voffset));
#undef KFN_STR
#undef KFN_STR2
#endif
}
#endif

// thread-local error text returned by kfn_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return KFN_OK;
  return fail(KFN_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define KFN_HIP(call)                                              \
  do {                                                             \
    int _rc = ::kfn::check_hip((call), #call);                     \
    if (_rc != KFN_OK) return _rc;                                 \
  } while (0)

#define KFN_LAUNCH_CHECK(name)                                     \
  do {                                                             \
    int _rc = ::kfn::check_hip(hipGetLastError(), name);           \
    if (_rc != KFN_OK) return _rc;                                 \
  } while (0)

#define KFN_REQUIRE(cond, ...)                                     \
  do {                                                             \
    if (!(cond)) return ::kfn::fail(KFN_ERR_ARG, __VA_ARGS__);     \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: `done`
// (one static per kernel instantiation) holds a bit per device it has been applied on, so a
// process that drives several GPUs (or host threads on different devices) sets it once on
// each.  Thread-safe: the attribute is idempotent, the bit set is atomic.
inline int set_max_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  KFN_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return KFN_OK;
  KFN_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.fetch_or(bit, std::memory_order_release);
  return KFN_OK;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Dynamic LDS (bytes per workgroup) of the kernels behind the Winograd entry points -- what kfn_winograd_lds_bytes()
// answers; each is defined beside the kernel it describes.  < 0: no such form in that file.
int wino_f43_lds_bytes(int wino_form);     // kfn_wino4.hip: KFN_WINO_FORM_F43_FOUR_WAVE / _EIGHT_WAVE
int wino_s2_lds_bytes(int wino_form, int operand_dtype);   // kfn_wino_s2.hip: AUTO (four waves) / KFN_WINO_FORM_S2_EIGHT_WAVE
int wino_s2c_lds_bytes();                  // kfn_wino_s2c.hip: KFN_WINO_FORM_S2_F42
int wino_s2c_supported(const kfn_conv_desc* d);   // (normalised descriptor)
int launch_wino_s2c(const kfn_conv_desc* d, const float* x, const void* u_packed, const float* bias, float* y, void* stream);
int wino_fused_lds_bytes(const kfn_conv_desc* d);   // kfn_wino3.hip: the form kfn_conv2d_winograd_fused routes `d` to
// y[pix][c] = relu(bias[c] + sum_s ws[s][pix][c]), s ascending (kfn_wino4.hip): the second launch of the split-K forms
int launch_splitk_reduce(const float* ws, int k_split, long pixels, const float* bias, int relu, float* y, int Cout, int ldy,
                         hipStream_t stream);
constexpr int WINO2_LDS_BYTES = 2 * 16384;          // wino2_kernel's two raw-patch buffers (kfn_wino2.hip asserts it)

// kfn_conv_desc crosses the ABI by pointer and grows at its end; `struct_size` (first member) says how many bytes the
// CALLER's object has.  Every entry point works on a full-size copy whose missing tail is zero (= AUTO / fp32
// defaults) and never touches the caller's memory beyond struct_size.
inline int conv_desc_in(const kfn_conv_desc* in, kfn_conv_desc* out, const char* who, bool layouts_ok = false) {
  if (in == nullptr) return fail(KFN_ERR_ARG, "%s: null descriptor", who);
  const int32_t sz = in->struct_size;
  const int32_t min_sz = (int32_t)(offsetof(kfn_conv_desc, config) + sizeof(int32_t));   // the first-round struct
  if (sz < min_sz || sz > (int32_t)sizeof(kfn_conv_desc) || (sz & 3) != 0)
    return fail(KFN_ERR_ARG,
                "%s: kfn_conv_desc.struct_size = %d (this library: %d bytes, minimum %d) -- set it to sizeof(kfn_conv_desc) "
                "(KFN_CONV_DESC_INIT); hosts built for ABI <= 4 (no struct_size member) must be rebuilt against ABI %d",
                who, (int)sz, (int)sizeof(kfn_conv_desc), (int)min_sz, KFN_ABI_VERSION);
  std::memset(out, 0, sizeof(*out));
  std::memcpy(out, in, (size_t)sz);
  out->struct_size = (int32_t)sizeof(kfn_conv_desc);
  if ((unsigned)out->x_layout > KFN_LAYOUT_C16 || (unsigned)out->y_layout > KFN_LAYOUT_C16)
    return fail(KFN_ERR_ARG, "%s: unknown activation layout (x_layout %d, y_layout %d)", who, out->x_layout, out->y_layout);
  if (!layouts_ok && (out->x_layout != KFN_LAYOUT_NHWC || out->y_layout != KFN_LAYOUT_NHWC))
    return fail(KFN_ERR_UNSUPPORTED, "%s reads and writes NHWC only (x_layout %d, y_layout %d): the channel-blocked layout exists "
                "for kfn_conv2d_winograd_f43 (eight-wave form) and kfn_conv2d_winograd_s2 (F(4,2) form)", who, out->x_layout, out->y_layout);
  return KFN_OK;
}
// (the declaration shadows the caller's pointer with the normalised copy)
#define KFN_CONV_DESC_IN(d, who)                                   \
  kfn_conv_desc d##_full;                                          \
  {                                                                \
    int _rc = ::kfn::conv_desc_in(d, &d##_full, who);              \
    if (_rc != KFN_OK) return _rc;                                 \
  }                                                                \
  d = &d##_full
// ... for the entry points that take KFN_LAYOUT_C16 activations
#define KFN_CONV_DESC_IN_LAYOUTS(d, who)                           \
  kfn_conv_desc d##_full;                                          \
  {                                                                \
    int _rc = ::kfn::conv_desc_in(d, &d##_full, who, true);        \
    if (_rc != KFN_OK) return _rc;                                 \
  }                                                                \
  d = &d##_full

// Wave-wide (64 lanes) sum / max on the VALU: four DPP butterfly steps leave every lane of a 16-lane row with the
// row's total (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- each step adds a lane's
// value to its partner's, so all lanes of a row evaluate the same tree), the four row totals are read into
// scalars and combined as (r0 + r1) + (r2 + r3).  ~12 instructions; six __shfl_xor steps are six dependent
// ds_bpermute round trips through the LDS crossbar (~100 cycles each).  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// TF 'SAME' padding rule (tf.layers.conv2d, cnn_wrapper/network.py:126-135):
// out = ceil(in/stride); pad_total = max((out-1)*stride + k - in, 0); before = total/2.
inline void same_pad(int in, int k, int stride, int* out, int* before) {
  *out = (in + stride - 1) / stride;
  int total = (*out - 1) * stride + k - in;
  if (total < 0) total = 0;
  *before = total / 2;
}

}  // namespace kfn
