#!/bin/bash
# dev: fast3's per-phase cycles from the tile-stamp diagnostic library (built HERE, side by side with the product one):
#   TBNN_BUILD_TAG=tstamps TBNN_EXTRA_FLAGS=-DTBNN_TILE_STAMPS python -m tensorbnn_amd.build
TBNN_LIB=$PWD/tensorbnn_amd/libtbnn_tstamps.so timeout 200 python3 ${TBNN_STAMP_TOOL:-tools/experiments/tilestamps.py}
