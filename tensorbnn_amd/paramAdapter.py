"""(eps, L) adapter with the reference's class name and signature
(tensorBNN/paramAdapter.py:11-292), running in host C++ inside libtbnn
(tbnn_adapter_*).  The reference's adapter is host-side code as well."""
import numpy as np

from . import _native as nat


class paramAdapter(object):
    def __init__(self, e1, L1, el, eu, eNumber, Ll, Lu, lStep, m, k, a=4, delta=0.1, cores=4, strikes=10,
                 randomSteps=10, seed=0):
        # `cores` is stored and never used by the reference (paramAdapter.py:90); `strikes` is
        # ignored there too (maxStrikes is hard-coded to 50, :92)
        self.cores = cores
        self.currentE, self.currentL = np.float32(e1), np.int32(L1)
        self._a = nat.Adapter(e1, L1, el, eu, eNumber, Ll, Lu, lStep, m, k, a=a, delta=delta,
                              randomSteps=randomSteps, seed=seed)
        self.lastSJD = None

    def update(self, state, inject_u=-1.0, inject_e=-1, inject_l=-1):
        """state: list of state tensors (as the reference passes) or one flat vector."""
        if isinstance(state, (list, tuple)):
            state = np.concatenate([np.asarray(s, dtype=np.float32).reshape(-1) for s in state])
        e, L, sjd = self._a.update(state, inject_u, inject_e, inject_l)
        self.currentE, self.currentL, self.lastSJD = np.float32(e), np.int32(L), sjd
        return self.currentE, self.currentL
