#!/bin/bash
# dev: wide-path tests + bench lines
python3 -m pytest tests/test_gpu_wide.py tests/test_gpu_jit.py tests/test_gpu_comm.py -x -q 2>&1 | tail -2
for wl in "$@"; do python3 bench.py --workload $wl --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"kernel_us": [0-9.]*\|"frac": [0-9.]*\|"accept_ratio": [0-9.]*' | tr '\n' ' '; echo " [$wl]"; done
