// Host interface of the wide-layer path (kernels_wide.hpp), compiled in its own
// translation unit (tbnn_wide.hip) so that the two big kernel families build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include "common.hpp"

struct WidePlan {
    int id = -1;
    int gridA = 0;               // k_chain_wide workgroups (= entries of pstat)
    int gridB = 0;               // k_dw_wide workgroups
    int wg_lo[TBNN_MAX_LAYERS + 1] = {0};
    size_t store_floats = 0;     // a_l / delta_l blocks
    size_t slabA_floats = 0;     // gridA x compact slab (first + last layer)
    size_t slabB_floats = 0;     // gridB x middle-layer slab
    int img_floats = 0;
    int fwd_ok = 0;              // a forward-only instantiation exists (wide_forward)
};

int wide_lookup(const NetDev& nd);
const char* wide_name(int id);
void wide_image_map_id(int id, int* map);          // 2P ints
void wide_plan(int id, long n, WidePlan& plan);
// k_chain_wide + k_dw_wide + k_reduce_wide on `st`; out: one dense gradient row of P floats
int wide_launch(const WidePlan& plan, hipStream_t st, const NetDev& nd, const float* qimg, const float* eta,
                const float* X, const float* Y, long n, float* store, float* slabA, float* slabB, double* pstat, float* out);
// forward only (network.predict): fout[d_out][n]; qimg = padded image of the weights
int wide_forward(int id, hipStream_t st, const NetDev& nd, const float* qimg, const float* X, long n, float* fout);
