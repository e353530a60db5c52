"""dev: shader-clock stamps of the cooperative tail tile of k_fwd_bwd_fast3 at configs[1] (wave 0 of workgroup 0), from the
tile-stamp diagnostic library:
diagnostic build (=2: outer phases only; the fine stamps serialise what they bracket):
  TBNN_BUILD_TAG=ts2 TBNN_EXTRA_FLAGS=-DTBNN_TILE_STAMPS=2 python -m tensorbnn_amd.build
  TBNN_LIB=$PWD/tensorbnn_amd/libtbnn_ts2.so python3 tools/experiments/coopstamps.py [dims, e.g. 8,50,50,1 -> run-time compiled:
  TBNN_JIT_FLAGS=-DTBNN_TILE_STAMPS=2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C, numpy as np
import tensorbnn_amd._native as nat
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [5, 50, 50, 50, 1], 100000)
ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
for _ in range(5): ch.logp_grad()
out = (C.c_uint64 * 64)()
if hasattr(nat.lib, "tbnn_debug_tile_stamps"):      # ahead-of-time instantiation (configs[1]): the diagnostic build of libtbnn
    nat.lib.tbnn_debug_tile_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    assert nat.lib.tbnn_debug_tile_stamps(ch._h, out) == 0
if not any(out):                                     # a run-time compiled kernel library: its own copy of the stamps
    from tensorbnn_amd import jit
    klib = C.CDLL(jit.build(layers, lik))
    klib.tbnn_jit_tile_stamps.argtypes = [C.POINTER(C.c_uint64)]
    assert klib.tbnn_jit_tile_stamps(out) == 0
t = np.array(list(out), dtype=np.float64)
seq = [(40, 'start'), (41, 'fwd0 own tile'), (42, 'fwd0 barrier'), (43, 'fwd0 read + image'), (44, 'fwd1 own tile'), (45, 'fwd1 barrier'),
       (46, 'fwd1 read + image'), (47, 'fwd2 own tile'), (48, 'fwd2 barrier'), (49, 'fwd2 read + image'), (25, 'last layer'), (26, 'lik + frDW'), (27, 'da(L)'),
       (50, 'dW2'), (51, 'delta2 -> dz1'), (52, 'dW1'), (53, 'delta1 -> dz0'), (54, 'dW0'), (56, 'end')]
fine = t[41] != 0          # TBNN_JIT_FLAGS=-DTBNN_TILE_STAMPS=2: only the outer phases (the fine stamps serialise what they bracket)
prev = t[40]
for k, name in (seq[1:] if fine else []):
    if t[k] == 0: continue
    print(f'{name:20s} {t[k] - prev:8.0f}'); prev = t[k]
print('total', t[56] - t[40])
# the kernel's phases around it (stamps 57..63: start, prologue end, first tile end, tile loop end, kernel end, coop end, staged)
print(f'prologue {t[58] - t[57]:.0f}, first tile {t[59] - t[58]:.0f}, other tiles {t[60] - t[59]:.0f}, tile loop end -> coop tile start {t[40] - t[60]:.0f}, '
      f'coop tile {t[56] - t[40]:.0f}, coop tile end -> coop end {t[62] - t[56]:.0f}, staging {t[63] - t[62]:.0f}, slabs + fringe out {t[61] - t[63]:.0f}; '
      f'whole launch {t[61] - t[57]:.0f}')
if t[35]: print(f'epilogue: tile sums {t[35] - t[63]:.0f}, fringe sums {t[36] - t[35]:.0f}, barrier {t[37] - t[36]:.0f}, stores + end {t[61] - t[37]:.0f}')
if fine: print(f'staging: stat sum {t[20] - t[62]:.0f}, barrier {t[21] - t[20]:.0f}, dW tiles {t[22] - t[21]:.0f}, merge {t[23] - t[22]:.0f}, fringe partials {t[24] - t[23]:.0f}, barrier {t[63] - t[24]:.0f}')
