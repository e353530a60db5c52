import ctypes as C, numpy as np, os, sys
import tensorbnn_amd._native as nat
# load the diagnostic build instead
dbg = C.CDLL(os.path.join(os.path.dirname(nat.__file__), 'libtbnn_dbg.so'))
for name, res, args in nat.SYMBOLS:
    fn = getattr(dbg, name); fn.restype = res; fn.argtypes = args
nat.lib = dbg
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5,50,50,50,1], 114688)
ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
for _ in range(3): ch.logp_grad()
out=(C.c_uint64*64)()
dbg.tbnn_debug_tile_stamps.argtypes=[C.c_void_p, C.POINTER(C.c_uint64)]
dbg.tbnn_debug_tile_stamps(ch._h, out)
t=np.array(list(out),dtype=np.float64)
seq=[(0,'start',0),(1,'fwd L0',8),(2,'fwd L1',56),(3,'fwd L2',56),(4,'fwd L3',14),(9,'lik',0),(19,'iss3',0),(21,'dA3',4),
     (16,'iss2',0),(17,'dW3',16),(18,'dA2',56),(13,'iss1',0),(14,'dW2',64),(15,'dA1',56),(10,'iss0',0),(11,'dW1',64),(12,'dW0',16)]
prev=t[0]; tot=0
for k,name,m in seq[1:]:
    d=t[k]-prev; prev=t[k]
    print(f'{name:8s} cycles {d:8.0f}  mfma {m:4d}  ideal {m*32:6d}')
print('total', t[12]-t[0], 'ideal', 410*32)
