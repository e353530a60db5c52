"""ctypes loader of the C restatement (oracle/c/tbnn_oracle.c) -- TEST
INFRASTRUCTURE ONLY (tests + bench.py's cpu_baseline leg)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libtbnn_oracle.so")
MAX_THREADS = None


class ONet(C.Structure):
    _fields_ = [("nl", C.c_int), ("in_", C.c_int * 16), ("out", C.c_int * 16), ("act", C.c_int * 16),
                ("prior", C.c_int * 16), ("lik", C.c_int), ("fixed_sd", C.c_float)]


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "c", "tbnn_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE])
    lib = C.CDLL(LIB)
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    lib.oracle_logp_grad.restype = C.c_double
    lib.oracle_logp_grad.argtypes = [C.POINTER(ONet), fp, fp, fp, fp, C.c_long, fp, dp]
    lib.oracle_hmc_step.restype = C.c_int
    lib.oracle_hmc_step.argtypes = [C.POINTER(ONet), fp, fp, fp, fp, C.c_long, C.c_float, C.c_int, fp, C.c_float,
                                    dp, dp, dp]
    lib.oracle_hmc_propose.restype = None
    lib.oracle_hmc_propose.argtypes = [C.POINTER(ONet), fp, fp, fp, fp, C.c_long, C.c_float, C.c_int, fp, fp, dp, dp, dp]
    lib.oracle_num_threads.restype = C.c_int
    lib.oracle_set_threads.restype = None
    lib.oracle_set_threads.argtypes = [C.c_int]
    global MAX_THREADS
    if MAX_THREADS is None:
        MAX_THREADS = lib.oracle_num_threads()       # before anyone calls oracle_set_threads (process-global in OpenMP)
    return lib


def make_net(spec):
    n = ONet()
    n.nl = len(spec.layers)
    for i, l in enumerate(spec.layers):
        n.in_[i], n.out[i], n.act[i], n.prior[i] = l.in_dim, l.out_dim, l.act, l.prior
    n.lik = spec.likelihood
    n.fixed_sd = spec.fixed_sd
    return n


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


class COracle:
    def __init__(self, spec, X, Y):
        self.lib = load()
        self.net = make_net(spec)
        self.X = np.ascontiguousarray(X, dtype=np.float32)
        self.Y = np.ascontiguousarray(Y, dtype=np.float32)
        self.n = self.X.shape[0]
        self.P = spec.n_params
        self.max_threads = MAX_THREADS
        # default: at most 32 threads (the GPU boxes show 256 logical CPUs but schedule a fraction of them: more threads
        # than that ran 10x slower there); bench.py scans the count itself
        self.set_threads(int(os.environ.get("TBNN_ORACLE_THREADS", min(MAX_THREADS, 32))))

    def set_threads(self, n):
        self.lib.oracle_set_threads(int(n))
        self.threads = self.lib.oracle_num_threads()

    def logp_grad(self, theta, eta):
        th = np.ascontiguousarray(theta, dtype=np.float32)
        et = np.ascontiguousarray(eta, dtype=np.float32)
        g = np.empty(self.P, dtype=np.float32)
        st = C.c_double()
        lp = self.lib.oracle_logp_grad(C.byref(self.net), _p(th), _p(et), _p(self.X), _p(self.Y), self.n, _p(g),
                                       C.byref(st))
        return lp, g, st.value

    def hmc_step(self, theta, eta, eps, L, p0, log_u):
        th = np.array(theta, dtype=np.float32)
        et = np.ascontiguousarray(eta, dtype=np.float32)
        p = np.ascontiguousarray(p0, dtype=np.float32)
        lar, lo, ln = C.c_double(), C.c_double(), C.c_double()
        acc = self.lib.oracle_hmc_step(C.byref(self.net), _p(th), _p(et), _p(self.X), _p(self.Y), self.n,
                                       float(eps), int(L), _p(p), float(log_u), C.byref(lar), C.byref(lo), C.byref(ln))
        return th, bool(acc), lar.value, lo.value, ln.value

    def hmc_propose(self, theta, eta, eps, L, p0):
        """(proposal q_L, log accept ratio, logp at theta, logp at q_L): the transition without the Metropolis decision"""
        th = np.ascontiguousarray(theta, dtype=np.float32)
        et = np.ascontiguousarray(eta, dtype=np.float32)
        p = np.ascontiguousarray(p0, dtype=np.float32)
        q = np.empty(self.P, dtype=np.float32)
        lar, lo, ln = C.c_double(), C.c_double(), C.c_double()
        self.lib.oracle_hmc_propose(C.byref(self.net), _p(th), _p(et), _p(self.X), _p(self.Y), self.n, float(eps), int(L),
                                    _p(p), _p(q), C.byref(lar), C.byref(lo), C.byref(ln))
        return q, lar.value, lo.value, ln.value
