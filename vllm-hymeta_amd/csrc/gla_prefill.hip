#include "common.h"
extern "C" int hymeta_gla_prefill_varlen(const hymeta_gla_prefill_args*, hymeta_stream_t) { return HYMETA_ERR_UNSUPPORTED; }
