// Included by a run-time generated translation unit (tensorbnn_amd/jit.py) and by the ahead-of-time registry (tbnn_tall.hip):
// the tall-fan-in fused kernel (kernels_tall.hpp).  It speaks the narrow family's launch interface (one gradient slab per
// workgroup).
#pragma once
#define TBNN_NO_FAST_REGISTRY
#include "kernels_tall.hpp"
#include "fused_ops.hpp"

template <class S>
struct JitTall {
    static int grid(long n) { return tall_grid<S>(n); }
    static int launch(int g, hipStream_t st, const NetDev* nd, const float* qimg, const float* eta, const float* X, const float* Y,
                      long n, float* slabs, int pitch, double* pstat, int nchains, ChainStride cs) {
        return tall_launch_t<S>(g, st, *nd, qimg, eta, X, Y, n, slabs, pitch, pstat, nchains, cs);
    }
    static int nforward(int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n, float* fouts,
                        long out_stride) {
        return tall_forward_t<S>(gx, nets, st, qimgs, img_stride, X, n, fouts, out_stride);
    }
    static void image_map(int* map) { tall_image_map<S>(map); }
    static void fill(FusedOps* o, const char* prefix = "jit-tall") {
        fused_ops_shape<S>(o, prefix);
        o->family = TBNN_FAMILY_NARROW;
        o->img_floats = TallCfg<S, TallPick<S>::NW>::IMG_FLOATS;
        o->image_map = &image_map; o->grid = &grid; o->launch = &launch; o->nforward = &nforward;
        o->plan = nullptr; o->wlaunch = nullptr; o->wforward = nullptr;
    }
};
