// Registry + launchers of the mid-width fused kernel (kernels_mid.hpp).
#include <hip/hip_runtime.h>
#define TBNN_NO_FAST_REGISTRY
#include "mid_api.hpp"
#include "kernels_mid.hpp"

using MShapeC5 = Shape<TBNN_ACT_RELU, TBNN_ACT_SIGMOID, true, 20, 100, 100, 2>;         // BASELINE configs[4]
using MShapeT1 = Shape<TBNN_ACT_TANH, TBNN_ACT_NONE, false, 4, 24, 40, 1>;              // test: ragged widths, one middle layer
using MShapeT2 = Shape<TBNN_ACT_SIGMOID, TBNN_ACT_SIGMOID, true, 20, 32, 48, 2>;        // test: widths % 16 == 0 (ones slot in its own tile)
using MShapeT3 = Shape<TBNN_ACT_RELU, TBNN_ACT_NONE, false, 7, 33, 18, 50, 2>;          // test: two middle layers

template <class S>
static bool mshape_matches(const NetDev& nd) {
    if (nd.nl != S::NL) return false;
    if ((nd.lik == TBNN_LIK_BERNOULLI) != S::BERN) return false;
    for (int l = 0; l < S::NL; ++l)
        if (nd.in[l] != S::D[l] || nd.out[l] != S::D[l + 1] || nd.act[l] != S::act(l)) return false;
    return true;
}

int mid_lookup(const NetDev& nd) {
    if (mshape_matches<MShapeC5>(nd)) return 0;
    if (mshape_matches<MShapeT1>(nd)) return 1;
    if (mshape_matches<MShapeT2>(nd)) return 2;
    if (mshape_matches<MShapeT3>(nd)) return 3;
    return -1;
}
const char* mid_name(int id) {
    switch (id) {
        case 0: return "mid<relu,sigmoid,bernoulli;20,100,100,2>";
        case 1: return "mid<tanh;4,24,40,1>";
        case 2: return "mid<sigmoid,sigmoid,bernoulli;20,32,48,2>";
        case 3: return "mid<relu;7,33,18,50,2>";
        default: return "mid<none>";
    }
}

#define MID_DISPATCH(id, CALL)                                    \
    switch (id) {                                                 \
        case 0: { using S = MShapeC5; CALL; } break;              \
        case 1: { using S = MShapeT1; CALL; } break;              \
        case 2: { using S = MShapeT2; CALL; } break;              \
        case 3: { using S = MShapeT3; CALL; } break;              \
        default: break;                                           \
    }

int mid_image_floats(int id) { int r = 0; MID_DISPATCH(id, r = MidCfg<S>::IMG_FLOATS); return r; }
void mid_image_map_id(int id, int* map) { MID_DISPATCH(id, mid_image_map<S>(map)); }
int mid_grid_id(int, long n) { return mid_grid(n); }
int mid_launch(int id, int grid, hipStream_t st, const NetDev& nd, const float* qimg, const float* eta, const float* X,
               const float* Y, long n, float* slabs, int pitch, double* pstat, int nchains, ChainStride cs) {
    int rc = -1;
    MID_DISPATCH(id, rc = mid_launch_t<S>(grid, st, nd, qimg, eta, X, Y, n, slabs, pitch, pstat, nchains, cs));
    return rc;
}
int mid_forward(int id, int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n,
                float* fouts, long out_stride) {
    int rc = -1;
    MID_DISPATCH(id, rc = mid_forward_t<S>(gx, nets, st, qimgs, img_stride, X, n, fouts, out_stride));
    return rc;
}

#ifdef MID_STAMPS
extern "C" int tbnn_debug_mid_stamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_mid_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
