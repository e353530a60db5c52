// Host interface of the mid-width fused kernel (kernels_mid.hpp), compiled in its own translation unit
// (tbnn_mid.hip) so that the kernel families build side by side.  Same launch signature as the narrow family:
// one gradient slab per workgroup, reduced by k_update.
#pragma once
#include <hip/hip_runtime.h>
#include "common.hpp"

int mid_lookup(const NetDev& nd);                  // -1: no ahead-of-time instantiation covers this network
const char* mid_name(int id);
int mid_image_floats(int id);
void mid_image_map_id(int id, int* map);           // 2P ints
int mid_grid_id(int id, long n);
int mid_launch(int id, int grid, hipStream_t st, const NetDev& nd, const float* qimg, const float* eta, const float* X,
               const float* Y, long n, float* slabs, int pitch, double* pstat, int nchains = 1, ChainStride cs = ChainStride{0, 0, 0, nullptr, 0});
// forward only: `nets` networks (grid.y; images img_stride floats apart), fouts[net][d_out][n]
int mid_forward(int id, int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n,
                float* fouts, long out_stride);
