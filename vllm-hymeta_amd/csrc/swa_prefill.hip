#include "common.h"
extern "C" int hymeta_meta_swa_prefill(const hymeta_swa_prefill_args*, hymeta_stream_t) { return HYMETA_ERR_UNSUPPORTED; }
