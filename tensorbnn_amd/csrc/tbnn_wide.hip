// Registry + launchers of the wide-layer kernels (kernels_wide.hpp).
#include <hip/hip_runtime.h>
#include <algorithm>
#define TBNN_NO_FAST_REGISTRY
#include "wide_api.hpp"
#include "kernels_wide.hpp"

using WShapeC4 = Shape<TBNN_ACT_RELU, TBNN_ACT_NONE, false, 10, 200, 200, 200, 1>;      // BASELINE configs[3]
using WShapeC5 = Shape<TBNN_ACT_RELU, TBNN_ACT_SIGMOID, true, 20, 100, 100, 2>;         // BASELINE configs[4]
using WShapeT1 = Shape<TBNN_ACT_TANH, TBNN_ACT_NONE, false, 3, 20, 36, 2>;              // test: ragged widths, one middle layer
using WShapeT2 = Shape<TBNN_ACT_SIGMOID, TBNN_ACT_SIGMOID, true, 20, 32, 16, 48, 2>;    // test: widths % 16 == 0 (ones slot in its own tile)

template <> struct WideForceStream<WShapeT2> { static constexpr bool value = true; };   // keeps the ring path under the small-shape tests

template <class S>
static bool wshape_matches(const NetDev& nd) {
    if (nd.nl != S::NL) return false;
    if ((nd.lik == TBNN_LIK_BERNOULLI) != S::BERN) return false;
    for (int l = 0; l < S::NL; ++l)
        if (nd.in[l] != S::D[l] || nd.out[l] != S::D[l + 1] || nd.act[l] != S::act(l)) return false;
    return true;
}

int wide_lookup(const NetDev& nd) {
    if (wshape_matches<WShapeC4>(nd)) return 0;
    if (wshape_matches<WShapeC5>(nd)) return 1;
    if (wshape_matches<WShapeT1>(nd)) return 2;
    if (wshape_matches<WShapeT2>(nd)) return 3;
    return -1;
}
const char* wide_name(int id) {
    switch (id) {
        case 0: return "wide<relu;10,200,200,200,1>";
        case 1: return "wide<relu,sigmoid,bernoulli;20,100,100,2>";
        case 2: return "wide<tanh;3,20,36,2>";
        case 3: return "wide<sigmoid,sigmoid,bernoulli;20,32,16,48,2>";
        default: return "wide<none>";
    }
}

#define WIDE_DISPATCH(id, CALL)                                   \
    switch (id) {                                                 \
        case 0: { using S = WShapeC4; CALL; } break;              \
        case 1: { using S = WShapeC5; CALL; } break;              \
        case 2: { using S = WShapeT1; CALL; } break;              \
        case 3: { using S = WShapeT2; CALL; } break;              \
        default: break;                                           \
    }

void wide_image_map_id(int id, int* map) { WIDE_DISPATCH(id, wide_image_map<S>(map)); }
void wide_plan(int id, long n, WidePlan& plan) { plan.id = id; WIDE_DISPATCH(id, (wide_plan_t<S>(n, plan), plan.fwd_ok = wide_forward_ok<S>() ? 1 : 0)); }
int wide_forward(int id, hipStream_t st, const NetDev& nd, const float* qimg, const float* X, long n, float* fout) {
    int rc = -1;
    WIDE_DISPATCH(id, rc = wide_forward_t<S>(st, nd, qimg, X, n, fout));
    return rc;
}
int wide_launch(const WidePlan& plan, hipStream_t st, const NetDev& nd, const float* qimg, const float* eta,
                const float* X, const float* Y, long n, float* store, float* slabA, float* slabB, double* pstat, float* out) {
    int rc = -1;
    WIDE_DISPATCH(plan.id, rc = wide_launch_t<S>(plan, st, nd, qimg, eta, X, Y, n, store, slabA, slabB, pstat, out));
    return rc;
}

#ifdef WIDE_STAMPS
extern "C" int tbnn_debug_wide_stamps(unsigned long long* out256) {
    return hipMemcpyFromSymbol(out256, HIP_SYMBOL(g_wide_stamps), 256 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
