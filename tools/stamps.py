import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5,50,50,50,1], 114688)
nat.lib.tbnn_debug_stamps.argtypes=[C.c_void_p, C.POINTER(C.c_uint64)]
for n in (16, 16384, 114688):
    ch = nat.Chain(layers, likelihood=lik); ch.set_data(X[:n], Y[:n]); ch.set_state(th); ch.set_hypers(eta)
    out=(C.c_uint64*16)()
    nat.lib.tbnn_debug_stamps(ch._h, out)
    t=np.array(list(out)[:8],dtype=np.float64); d=(t-t[0])*0.01; c=np.array(list(out)[8:13],dtype=np.float64); print("  clock MHz", (c[4]-c[0])/(d[4]+1e-9))
    print('n',n,'us: prologue',d[1],'first tile end',d[2],'loop end',d[3],'end',d[4], 'per-wave loop end', [(out[12+w]-out[0])*0.01 for w in range(4)], 'staged', d[5], 'slabout', d[6])
    ch.close()
