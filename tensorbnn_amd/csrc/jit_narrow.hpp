// Included by a run-time generated translation unit (tensorbnn_amd/jit.py) after `using S = Shape<...>;`
// with JIT_FAST3 = 1 (k_fwd_bwd_fast3) or 0 (k_fwd_bwd_fast).
#pragma once
#define TBNN_NO_FAST_REGISTRY
#include "kernels_fast3.hpp"
#include "kernels_traj.hpp"
#include "fused_ops.hpp"

template <class S, bool F3>
struct JitNarrow {
    static int grid(long n) {
        const long ntiles = (n + 15) / 16, wgs = (ntiles + FAST_WAVES - 1) / FAST_WAVES;
        return (int)(wgs < 256 ? wgs : 256);
    }
    static int launch(int g, hipStream_t st, const NetDev* nd, const float* qimg, const float* eta, const float* X, const float* Y,
                      long n, float* slabs, int pitch, double* pstat, int nchains, ChainStride cs) {
        if constexpr (F3)
            hipLaunchKernelGGL(k_fwd_bwd_fast3<S>, dim3(g, nchains), dim3(FAST_THREADS), 0, st, *nd, qimg, eta, X, Y, n, slabs, pitch, pstat,
                               (unsigned long long*)nullptr, cs);
        else
            hipLaunchKernelGGL(k_fwd_bwd_fast<S>, dim3(g, nchains), dim3(FAST_THREADS), 0, st, *nd, qimg, eta, X, Y, n, slabs, pitch, pstat,
                               (unsigned long long*)nullptr, cs);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    static int nforward(int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n, float* fouts,
                        long out_stride) {
        if constexpr (F3) {
            hipLaunchKernelGGL(k_forward_fast3<S>, dim3(gx, nets), dim3(FAST_THREADS), 0, st, qimgs, img_stride, X, n, fouts, out_stride);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        } else {
            return -1;
        }
    }
    // whole trajectories of small problems (kernels_traj.hpp): 16 waves where the images of 16 waves fit, else 4, else none
    // (-DTBNN_TRAJ_WAVES=0: none -- jit.py's second attempt when only the trajectory kernel of a shape needs scratch memory)
    static constexpr int TRAJ_NW = (!F3 || TBNN_TRAJ_WAVES == 0) ? 0 : (TBNN_TRAJ_WAVES == 16 && TrajCfg<S, 16>::OK) ? 16 : (TrajCfg<S, 4>::OK ? 4 : 0);
    static int traj(int nchains, hipStream_t st, const NetDev* nd, const float* qimg, long img_stride, const float* eta, const float* X, const float* Y, long n,
                    float* q, float* p, float* g, float* gd, const int* imgmap, double* pstat, int nstat, float eps, int L, const StepCtl* ctl) {
        if constexpr (TRAJ_NW > 0) {
            hipLaunchKernelGGL((k_traj_fast3<S, TRAJ_NW>), dim3(1, nchains), dim3(64 * TRAJ_NW), 0, st, *nd, qimg, img_stride, eta, X, Y, n, q, p, g, gd, imgmap,
                               pstat, nstat, eps, L, ctl);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        } else {
            return -1;
        }
    }
    static void image_map(int* map) { ImageMap<S, 0>::run(map); }
    static void fill(FusedOps* o) {
        fused_ops_shape<S>(o, F3 ? "jit-fast3" : "jit-fast");
        o->family = TBNN_FAMILY_NARROW;
        o->img_floats = FastCfg<S>::STATIC_FLOATS;
        o->image_map = &image_map; o->grid = &grid; o->launch = &launch;
        o->nforward = F3 ? &nforward : nullptr;
        o->plan = nullptr; o->wlaunch = nullptr; o->wforward = nullptr;
        o->traj = TRAJ_NW > 0 ? &traj : nullptr;
        o->traj_max_rows = TRAJ_NW == 16 ? TBNN_TRAJ_MAX_ROWS : TRAJ_NW == 4 ? TBNN_TRAJ_MAX_ROWS_4 : 0;
    }
};

#ifdef TBNN_TILE_STAMPS
// diagnostic build (TBNN_JIT_FLAGS=-DTBNN_TILE_STAMPS): this library's copy of the shader-clock stamps (tools/experiments/coopstamps.py)
extern "C" int tbnn_jit_tile_stamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_tile_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
