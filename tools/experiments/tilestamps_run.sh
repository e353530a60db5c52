#!/bin/bash
# dev: build the tile-stamp diagnostic library next to the product one and print fast3's per-phase cycles
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form -DTBNN_TILE_STAMPS $TBNN_STAMP_FLAGS -Wno-unused-value -o tensorbnn_amd/libtbnn_dbg.so tensorbnn_amd/csrc/tbnn_api.hip tensorbnn_amd/csrc/tbnn_wide.hip tensorbnn_amd/csrc/adapter.cpp 2>&1 | grep -E "error" 
timeout 200 python3 ${TBNN_STAMP_TOOL:-tools/experiments/tilestamps.py}
rm -f tensorbnn_amd/libtbnn_dbg.so
