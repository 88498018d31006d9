// Shared helpers for the libkfnet_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <atomic>
#include <cstdint>
#include "../../include/kfnet_hip.h"

namespace kfn {

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

// TF 'SAME' padding rule (tf.layers.conv2d, cnn_wrapper/network.py:126-135):
// out = ceil(in/stride); pad_total = max((out-1)*stride + k - in, 0); before = total/2.
inline void same_pad(int in, int k, int stride, int* out, int* before) {
  *out = (in + stride - 1) / stride;
  int total = (*out - 1) * stride + k - in;
  if (total < 0) total = 0;
  *before = total / 2;
}

}  // namespace kfn
