import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C, numpy as np, os, sys
import tensorbnn_amd._native as nat
# the diagnostic build: TBNN_BUILD_TAG=tstamps TBNN_EXTRA_FLAGS=-DTBNN_TILE_STAMPS python -m tensorbnn_amd.build, then
# TBNN_LIB=tensorbnn_amd/libtbnn_tstamps.so python tools/experiments/tilestamps.py
dbg = nat.lib
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5,50,50,50,1], 114688)
ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
for _ in range(3): ch.logp_grad()
out=(C.c_uint64*64)()
dbg.tbnn_debug_tile_stamps.argtypes=[C.c_void_p, C.POINTER(C.c_uint64)]
dbg.tbnn_debug_tile_stamps(ch._h, out)
t=np.array(list(out),dtype=np.float64)
seq=[(0,'start',0),(1,'fwd L0',6),(2,'fwd L1',39),(3,'fwd L2',39),(4,'fwd VL',0),(30,'lik',0),(31,'frDW3+dA3',0),(32,'iss2',0),(33,'frDW2',0),(34,'dA2',39),
     (14,'iss1',0),(15,'dW2',48),(16,'frDW1',0),(17,'dA1',39),(10,'iss0',0),(11,'dW1',48),(12,'frDW0',0),(13,'dW0',12)]
prev=t[0]
for k,name,m in seq[1:]:
    d=t[k]-prev; prev=t[k]
    print(f'{name:10s} cycles {d:8.0f}  mfma {m:4d}  ideal {m*32:6d}')
print('total', t[13]-t[0], 'ideal', 270*32)
