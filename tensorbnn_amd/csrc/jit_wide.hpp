// Included by a run-time generated translation unit (tensorbnn_amd/jit.py): the wide-layer family.
#pragma once
#define TBNN_NO_FAST_REGISTRY
#include "wide_api.hpp"
#include "kernels_wide.hpp"
#include "fused_ops.hpp"

template <class S>
struct JitWide {
    static void plan(long n, WidePlan* p) { p->id = -2; wide_plan_t<S>(n, *p); }
    static int wlaunch(const WidePlan* p, hipStream_t st, const NetDev* nd, const float* qimg, const float* eta, const float* X,
                       const float* Y, long n, float* store, float* slabA, float* slabB, double* pstat, float* out) {
        return wide_launch_t<S>(*p, st, *nd, qimg, eta, X, Y, n, store, slabA, slabB, pstat, out);
    }
    static int wforward(hipStream_t st, const NetDev* nd, const float* qimg, const float* X, long n, float* fout) {
        return wide_forward_t<S>(st, *nd, qimg, X, n, fout);
    }
    static void image_map(int* map) { wide_image_map<S>(map); }
    static void fill(FusedOps* o) {
        fused_ops_shape<S>(o, WideCfg<S>::RESIDENT ? "jit-wide(resident)" : "jit-wide");
        o->family = TBNN_FAMILY_WIDE;
        o->img_floats = WideCfg<S>::IMG_FLOATS;
        o->image_map = &image_map; o->grid = nullptr; o->launch = nullptr;
        o->plan = &plan; o->wlaunch = &wlaunch;
        o->wforward = wide_forward_ok<S>() ? &wforward : nullptr;
        o->nforward = nullptr;
    }
};
