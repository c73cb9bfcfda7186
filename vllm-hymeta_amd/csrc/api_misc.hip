// Entry points that launch nothing: ABI version and error names.
#include "common.h"

extern "C" int hymeta_abi_version(void) { return HYMETA_ABI_VERSION; }

extern "C" const char* hymeta_strerror(int code) {
  switch (code) {
    case HYMETA_OK: return "ok";
    case HYMETA_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or inconsistent shape)";
    case HYMETA_ERR_UNSUPPORTED: return "unsupported shape, dtype, stride or mode";
    case HYMETA_ERR_WORKSPACE: return "workspace missing or too small";
    case HYMETA_ERR_LAUNCH: return "kernel launch failed";
    case HYMETA_ERR_SCRATCH_LIMIT: return "problem needs more scratch than the allocation-free entry point holds (HYMETA_STATIC_SCRATCH_BYTES)";
    default: return "unknown hymeta status";
  }
}

extern "C" int hymeta_stream_capture_id(hymeta_stream_t stream, uint64_t* id) {
  if (id == nullptr) return HYMETA_ERR_INVALID_ARG;
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  if (hipStreamGetCaptureInfo(static_cast<hipStream_t>(stream), &status, &cid) != hipSuccess) {
    (void)hipGetLastError();
    return HYMETA_ERR_LAUNCH;
  }
  *id = status == hipStreamCaptureStatusActive ? static_cast<uint64_t>(cid) : 0;
  return HYMETA_OK;
}
