// Table of one shape-specialised implementation of the fused forward+backward pass: what a run-time
// compiled kernel library (tensorbnn_amd/jit.py -> tbnn_register_kernel_lib) hands to libtbnn.
// Plain C layout: the table crosses a dlopen boundary.
#pragma once
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "wide_api.hpp"

#define TBNN_JIT_ABI 6      // 5: ChainStride carries the per-chain step control; 6: the trajectory kernel of small problems (kernels_traj.hpp)
enum { TBNN_FAMILY_NARROW = 1, TBNN_FAMILY_WIDE = 2 };

struct FusedOps {
    int abi;                                  // TBNN_JIT_ABI
    int family;                               // TBNN_FAMILY_*
    int nl, dims[TBNN_MAX_LAYERS + 1];        // the shape the kernels were instantiated for
    int hact, lact, bern;
    char name[128];
    int img_floats;                           // padded weight image (k_update scatters theta into it through image_map)
    void (*image_map)(int* map /* 2P */);
    // narrow family: one kernel, one gradient slab per workgroup
    int (*grid)(long n);
    // nchains / cs: gridDim.y = chains of a multi-chain handle, their image / eta / slab strides (1 and zeros for one chain)
    int (*launch)(int grid, hipStream_t st, const NetDev* nd, const float* qimg, const float* eta, const float* X,
                  const float* Y, long n, float* slabs, int pitch, double* pstat, int nchains, ChainStride cs);
    // narrow family, optional: forward only for `nets` networks (images img_stride floats apart), fout[net][d_out][n]
    int (*nforward)(int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n, float* fouts,
                    long out_stride);
    // narrow family, optional: the L leapfrog steps of a transition in one launch, one workgroup per chain, for problems of at most
    // traj_max_rows rows (kernels_traj.hpp; null / 0: none -- the per-step kernels)
    int traj_max_rows;
    int (*traj)(int nchains, hipStream_t st, const NetDev* nd, const float* qimg, long img_stride, const float* eta, const float* X, const float* Y, long n,
                float* q, float* p, float* g, float* gd, const int* imgmap, double* pstat, int nstat, float eps, int L, const StepCtl* ctl);
    // wide family: k_chain_wide + k_dw_wide + k_reduce_wide
    void (*plan)(long n, WidePlan* plan);
    int (*wlaunch)(const WidePlan* plan, hipStream_t st, const NetDev* nd, const float* qimg, const float* eta,
                   const float* X, const float* Y, long n, float* store, float* slabA, float* slabB, double* pstat, float* out);
    // wide family, optional: forward only, fout[d_out][n] (null: the generic forward kernel is used)
    int (*wforward)(hipStream_t st, const NetDev* nd, const float* qimg, const float* X, long n, float* fout);
};

static inline bool fused_ops_match(const FusedOps& o, const NetDev& nd) {
    if (o.nl != nd.nl || (nd.lik == TBNN_LIK_BERNOULLI) != (o.bern != 0)) return false;
    for (int l = 0; l < nd.nl; ++l) {
        if (nd.in[l] != o.dims[l] || nd.out[l] != o.dims[l + 1]) return false;
        if (nd.act[l] != (l == nd.nl - 1 ? o.lact : ((o.hact & TBNN_ACT_PACKED) ? (o.hact >> (3 * l)) & 7 : o.hact))) return false;
    }
    return true;
}

template <class S>
static inline void fused_ops_shape(FusedOps* o, const char* prefix) {
    o->abi = TBNN_JIT_ABI;
    o->nl = S::NL;
    for (int i = 0; i <= S::NL; ++i) o->dims[i] = S::D[i];
    o->hact = S::HCODE; o->lact = S::LACT; o->bern = S::BERN ? 1 : 0;      // (hact: one activation, or the packed per-layer code: Shape)
    static const char* an[] = {"none", "relu", "tanh", "sigmoid", "exp", "elu", "?", "?"};
    int k = snprintf(o->name, sizeof(o->name), "%s<", prefix);
    if (S::HCODE & TBNN_ACT_PACKED) {
        for (int l = 0; l + 1 < S::NL && k < (int)sizeof(o->name) - 16; ++l) k += snprintf(o->name + k, sizeof(o->name) - k, l ? "+%s" : "%s", an[S::act(l) & 7]);
    } else k += snprintf(o->name + k, sizeof(o->name) - k, "%s", an[S::HCODE & 7]);
    k += snprintf(o->name + k, sizeof(o->name) - k, ",%s%s;", an[S::LACT], S::BERN ? ",bernoulli" : "");
    for (int i = 0; i <= S::NL && k < (int)sizeof(o->name) - 8; ++i) k += snprintf(o->name + k, sizeof(o->name) - k, i ? ",%d" : "%d", S::D[i]);
    snprintf(o->name + k, sizeof(o->name) - k, ">");
}
