// placeholder, replaced below
#include "../../include/tbnn.h"
extern "C" int tbnn_adapter_create(float, int32_t, float, float, int32_t, int32_t, int32_t, int32_t, int32_t, double, float, float, int32_t, uint64_t, tbnn_adapter_handle* out) { if (out) *out = nullptr; return -9; }
extern "C" int tbnn_adapter_destroy(tbnn_adapter_handle) { return 0; }
extern "C" int tbnn_adapter_update(tbnn_adapter_handle, const float*, int32_t, float, int32_t, int32_t, float*, int32_t*, float*) { return -9; }
