// Host interface of the tall-fan-in fused kernel (kernels_tall.hpp), compiled in its own translation unit (tbnn_tall.hip):
// the ahead-of-time instantiations are FusedOps tables, the same interface a run-time compiled kernel library registers.
#pragma once
#include "common.hpp"
#include "fused_ops.hpp"

const FusedOps* tall_find(const NetDev& nd);       // null: no ahead-of-time instantiation covers this network
