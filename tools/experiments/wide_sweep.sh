#!/bin/bash
# round 2: build variants of the wide translation unit on the GPU box and time configs[3] / configs[4] (tools/widetime.py)
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; env $2 python -m tensorbnn_amd.build --force > /dev/null 2>&1 || { echo build failed; return; }; python tools/widetime.py c4 4 2>&1 | tail -1 | cut -c1-110; python tools/widetime.py c5 10 2>&1 | tail -1 | cut -c1-110; }
run "default" "X=1"
run "PD=1" "TBNN_WIDE_FLAGS=-DWIDE_PD=1"
run "PD=3" "TBNN_WIDE_FLAGS=-DWIDE_PD=3"
run "agpr-form" "TBNN_WIDE_AGPR_FORM=1"
run "agpr-form, dw0 builtin" "TBNN_WIDE_AGPR_FORM=1 TBNN_WIDE_FLAGS=-DWIDE_DW0_AGPR=0"
run "aprefetch" "TBNN_WIDE_FLAGS=-DWIDE_APREFETCH"
run "dw occ 1" "TBNN_WIDE_FLAGS=-DWIDE_DW_OCC_MAX=1"
run "dw pd 2" "TBNN_WIDE_FLAGS=-DWIDE_DW_PD=2"
