// Included by a run-time generated translation unit (tensorbnn_amd/jit.py): the mid-width fused kernel
// (kernels_mid.hpp).  It speaks the narrow family's launch interface (one gradient slab per workgroup).
#pragma once
#define TBNN_NO_FAST_REGISTRY
#include "kernels_mid.hpp"
#include "fused_ops.hpp"

template <class S>
struct JitMid {
    static int grid(long n) { return mid_grid(n); }
    static int launch(int g, hipStream_t st, const NetDev* nd, const float* qimg, const float* eta, const float* X, const float* Y,
                      long n, float* slabs, int pitch, double* pstat, int nchains, ChainStride cs) {
        return mid_launch_t<S>(g, st, *nd, qimg, eta, X, Y, n, slabs, pitch, pstat, nchains, cs);
    }
    static int nforward(int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n, float* fouts,
                        long out_stride) {
        return mid_forward_t<S>(gx, nets, st, qimgs, img_stride, X, n, fouts, out_stride);
    }
    static void image_map(int* map) { mid_image_map<S>(map); }
    static void fill(FusedOps* o) {
        fused_ops_shape<S>(o, "jit-mid");
        o->family = TBNN_FAMILY_NARROW;
        o->img_floats = MidCfg<S>::IMG_FLOATS;
        o->image_map = &image_map; o->grid = &grid; o->launch = &launch; o->nforward = &nforward;
        o->plan = nullptr; o->wlaunch = nullptr; o->wforward = nullptr;
    }
};
