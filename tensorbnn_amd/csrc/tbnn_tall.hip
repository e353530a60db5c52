// Registry of the ahead-of-time instantiations of the tall-fan-in fused kernel (kernels_tall.hpp).
#include <hip/hip_runtime.h>
#include <mutex>
#include "jit_tall.hpp"
#include "tall_api.hpp"

using TShapeMnist = Shape<TBNN_ACT_RELU, TBNN_ACT_SIGMOID, true, 784, 20, 20, 1>;     // docs/ClassificationExample.md:103-173
using TShapeT1 = Shape<TBNN_ACT_TANH, TBNN_ACT_NONE, false, 70, 24, 40, 1>;           // test: fan-in % 4 != 0 (scalar row loads), ragged widths
using TShapeT2 = Shape<TBNN_ACT_RELU, TBNN_ACT_NONE, false, 128, 16, 2>;              // test: no middle layer, ones slot opens a tile, two outputs
using TShapeT3 = Shape<TBNN_ACT_SIGMOID, TBNN_ACT_SIGMOID, true, 200, 33, 18, 50, 2>; // test: two middle layers, three M tiles in layer 0

static FusedOps g_tall[4];
static std::once_flag g_tall_once;

const FusedOps* tall_find(const NetDev& nd) {
    std::call_once(g_tall_once, [] {
        JitTall<TShapeMnist>::fill(&g_tall[0], "tall");
        JitTall<TShapeT1>::fill(&g_tall[1], "tall");
        JitTall<TShapeT2>::fill(&g_tall[2], "tall");
        JitTall<TShapeT3>::fill(&g_tall[3], "tall");
    });
    for (const FusedOps& o : g_tall) if (fused_ops_match(o, nd)) return &o;
    return nullptr;
}

#ifdef TBNN_TILE_STAMPS
// diagnostic build only: the stamps the last launch of a tall kernel left (kernels_tall.hpp: TALL_STAMP)
extern "C" int tbnn_tall_debug_stamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_tile_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
