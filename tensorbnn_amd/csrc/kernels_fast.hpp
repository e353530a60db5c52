// Shape-specialised fused forward+backward kernels (MFMA).  Registry stub:
// filled in by kernels_fast_impl.
#pragma once
#include "common.hpp"
static inline int fast_lookup(const NetDev&) { return -1; }
static inline const char* fast_name(int) { return "fast<none>"; }
static inline int fast_grid(int, long) { return 0; }
static inline int fast_launch(int, int, hipStream_t, const NetDev&, const float*, const float*, const float*,
                              const float*, long, float*, double*) { return -1; }
