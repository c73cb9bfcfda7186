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
    default: return "unknown hymeta status";
  }
}
