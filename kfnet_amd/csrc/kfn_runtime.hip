// kfn_runtime.hip -- error reporting, device facts and the thin allocator/stream/event
// plumbing of the C ABI (for hosts that do not bring PyTorch).
#include "kfn_common.h"

namespace kfn {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace kfn

extern "C" {

const char* kfn_last_error(void) { return kfn::err_buf(); }

int kfn_abi_version(void) { return KFN_ABI_VERSION; }

int kfn_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len) {
  hipDeviceProp_t prop;
  KFN_HIP(hipGetDeviceProperties(&prop, device));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return KFN_OK;
}

int kfn_malloc(void** dptr, size_t bytes) {
  KFN_REQUIRE(dptr, "kfn_malloc: null out pointer");
  KFN_HIP(hipMalloc(dptr, bytes));
  return KFN_OK;
}

int kfn_free(void* dptr) {
  KFN_HIP(hipFree(dptr));
  return KFN_OK;
}

int kfn_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  KFN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return KFN_OK;
}

int kfn_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  KFN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return KFN_OK;
}

int kfn_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  KFN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return KFN_OK;
}

int kfn_memset(void* dst, int value, size_t bytes, void* stream) {
  KFN_HIP(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return KFN_OK;
}

int kfn_stream_create(void** stream) {
  KFN_REQUIRE(stream, "kfn_stream_create: null out pointer");
  hipStream_t s;
  KFN_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return KFN_OK;
}

int kfn_stream_destroy(void* stream) {
  KFN_HIP(hipStreamDestroy((hipStream_t)stream));
  return KFN_OK;
}

int kfn_stream_sync(void* stream) {
  KFN_HIP(hipStreamSynchronize((hipStream_t)stream));
  return KFN_OK;
}

int kfn_event_create(void** event) {
  KFN_REQUIRE(event, "kfn_event_create: null out pointer");
  hipEvent_t e;
  KFN_HIP(hipEventCreate(&e));
  *event = e;
  return KFN_OK;
}

int kfn_event_destroy(void* event) {
  KFN_HIP(hipEventDestroy((hipEvent_t)event));
  return KFN_OK;
}

int kfn_event_record(void* event, void* stream) {
  KFN_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return KFN_OK;
}

int kfn_event_elapsed_ms(void* start, void* stop, float* ms) {
  KFN_REQUIRE(ms, "kfn_event_elapsed_ms: null out pointer");
  KFN_HIP(hipEventSynchronize((hipEvent_t)stop));
  KFN_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return KFN_OK;
}

}  // extern "C"
