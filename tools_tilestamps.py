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
names={0:'start',1:'fwd L0',2:'fwd L1',3:'fwd L2',4:'fwd L3',9:'lik',19:'b3 img',20:'b3 dA',21:'b3 dW',16:'b2 img',17:'b2 dA',18:'b2 dW',13:'b1 img',14:'b1 dA',15:'b1 dW',10:'b0 img',11:'b0 dA',12:'b0 dW'}
order=[0,1,2,3,4,9,19,20,21,16,17,18,13,14,15,10,11,12]
mf={1:8,2:56,3:56,4:14,9:0,19:0,20:4,21:16,16:0,17:56,18:64,13:0,14:56,15:64,10:0,11:0,12:16}
pair = os.environ.get('TBNN_FAST_SINGLE','0')!='1'
prev=t[0]
for k in order[1:]:
    d=t[k]-prev; prev=t[k]
    m=mf[k]*(2 if pair else 1)
    print(f'{names[k]:8s} cycles {d:8.0f}  mfma {m:4d}  ideal {m*32:6d}  eff {m*32/max(d,1):.2f}')
print('total', t[12]-t[0], 'ideal', (820 if pair else 410)*32)
