#!/bin/bash
TBNN_EXTRA_FLAGS="-DWIDE_STAMPS" python3 -m tensorbnn_amd.build --force > /dev/null 2>&1
timeout 200 python3 tools/widestamps.py ${1:-c4}
python3 -m tensorbnn_amd.build --force > /dev/null 2>&1
