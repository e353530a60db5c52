"""ctypes binding of libtbnn (include/tbnn.h) -- the only route to the HMC path.

There is no CPU fallback: if ``libtbnn.so`` is missing this module raises at
import time, and ``Chain`` raises when no gfx950 device is visible.  The
binding mirrors the header one to one; ``Chain`` is a thin object wrapper used
by ``tensorbnn_amd.network``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TBNN_LIB: a diagnostic build of the library next to the product one (tools/: stamped / variant builds)
LIB_PATH = os.environ.get("TBNN_LIB") or os.path.join(_HERE, "libtbnn.so")

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_EXP, ACT_ELU = 0, 1, 2, 3, 4, 5
PRIOR_CAUCHY, PRIOR_GAUSSIAN = 0, 1
LIK_GAUSSIAN, LIK_FIXED_GAUSSIAN, LIK_BERNOULLI = 0, 1, 2
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_FAST = 0, 1, 2
MAX_LAYERS = 16
ABI_VERSION = 3          # TBNN_ABI_VERSION of include/tbnn.h


class TbnnError(RuntimeError):
    pass


class LayerDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32), ("act", C.c_int32), ("prior", C.c_int32)]


class NetDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("layers", C.POINTER(LayerDesc)), ("likelihood", C.c_int32),
                ("fixed_sd", C.c_float), ("kernel", C.c_int32), ("reserved", C.c_int32)]


class StepOut(C.Structure):
    _fields_ = [("accepted", C.c_int32), ("n_leapfrog", C.c_int32), ("log_accept_ratio", C.c_float),
                ("accept_prob", C.c_float), ("logp_old", C.c_double), ("logp_new", C.c_double),
                ("kinetic_old", C.c_double), ("kinetic_new", C.c_double), ("sjd", C.c_double),
                ("device_us", C.c_float), ("fwdbwd_us", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/tbnn.h declares: (name, restype, argtypes)
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_H = C.c_void_p
SYMBOLS = [
    ("tbnn_last_error", C.c_char_p, []),
    ("tbnn_abi_version", C.c_int, []),
    ("tbnn_build_id", C.c_char_p, []),
    ("tbnn_lint_status", C.c_char_p, []),
    ("tbnn_device_count", C.c_int, []),
    ("tbnn_create", C.c_int, [C.POINTER(NetDesc), C.c_int, C.c_uint64, C.c_uint32, C.POINTER(_H)]),
    ("tbnn_create_multi", C.c_int, [C.POINTER(NetDesc), C.c_int, C.c_uint64, C.c_uint32, C.c_int32, C.POINTER(_H)]),
    ("tbnn_chain_count", C.c_int, [_H]),
    ("tbnn_destroy", C.c_int, [_H]),
    ("tbnn_param_count", C.c_int, [_H]),
    ("tbnn_hyper_count", C.c_int, [_H]),
    ("tbnn_kernel_name", C.c_char_p, [_H]),
    ("tbnn_last_transition_path", C.c_char_p, [_H]),
    ("tbnn_set_data", C.c_int, [_H, _fp, _fp, C.c_int64]),
    ("tbnn_set_data_device", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int64]),
    ("tbnn_set_state", C.c_int, [_H, _fp]),
    ("tbnn_get_state", C.c_int, [_H, _fp]),
    ("tbnn_set_hypers", C.c_int, [_H, _fp]),
    ("tbnn_get_hypers", C.c_int, [_H, _fp]),
    ("tbnn_logp_grad", C.c_int, [_H, _fp, _fp, _dp, _fp, _dp]),
    ("tbnn_forward", C.c_int, [_H, _fp, _fp, C.c_int64, _fp]),
    ("tbnn_hmc_step", C.c_int, [_H, C.c_float, C.c_int32, _fp, _fp, C.POINTER(StepOut), _dp]),
    ("tbnn_hmc_run", C.c_int, [_H, C.c_float, C.c_int32, C.c_int32, C.POINTER(StepOut)]),
    ("tbnn_hmc_step_each", C.c_int, [_H, _fp, C.POINTER(C.c_int32), C.POINTER(StepOut)]),
    ("tbnn_hmc_run_each", C.c_int, [_H, _fp, C.POINTER(C.c_int32), C.c_int32, C.POINTER(StepOut)]),
    ("tbnn_hyper_step_each", C.c_int, [_H, _fp, C.c_int32, C.POINTER(StepOut)]),
    ("tbnn_hyper_step", C.c_int, [_H, C.c_float, C.c_int32, _fp, _fp, C.POINTER(StepOut)]),
    ("tbnn_hyper_logp_grad", C.c_int, [_H, _fp, _dp, _fp]),
    ("tbnn_export_sample_device", C.c_int, [_H, C.c_void_p]),
    ("tbnn_debug_draw", C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_int32, _fp, _fp]),
    ("tbnn_set_epoch", C.c_int, [_H, C.c_uint32]),
    ("tbnn_set_profiling", C.c_int, [_H, C.c_int]),
    ("tbnn_debug_stamps", C.c_int, [_H, C.POINTER(C.c_uint64)]),
    ("tbnn_debug_fused_burst", C.c_int, [_H, C.c_int32, C.POINTER(C.c_float)]),
    ("tbnn_debug_momentum", C.c_int, [_H, _fp]),
    ("tbnn_set_validation", C.c_int, [_H, _fp, _fp, C.c_int64]),
    ("tbnn_predict", C.c_int, [_H, C.c_int, _fp, _fp]),
    ("tbnn_forward_many", C.c_int, [_H, _fp, C.c_int32, C.c_int64, C.c_int, _fp, C.c_int64, _fp]),
    ("tbnn_metrics", C.c_int, [_H, C.c_int, _fp, C.c_float, C.c_float, C.c_int, C.c_int, _dp]),
    ("tbnn_hyper_probs_many", C.c_int, [_H, C.POINTER(C.c_int32), _fp, C.c_int64, _fp, C.c_int64, C.c_int32, _dp]),
    ("tbnn_register_kernel_lib", C.c_int, [C.c_char_p]),
    ("tbnn_fused_kernel_available", C.c_int, [C.POINTER(NetDesc)]),
    ("tbnn_comm_unique_id", C.c_int, [C.POINTER(C.c_ubyte)]),
    ("tbnn_comm_create", C.c_int, [_H, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.POINTER(_H)]),
    ("tbnn_comm_count", C.c_int, [_H]),
    ("tbnn_comm_destroy", C.c_int, [_H]),
    ("tbnn_gather_samples", C.c_int, [_H, _H, C.c_void_p, _fp]),
    ("tbnn_set_row_shard", C.c_int, [_H, _H, C.c_int64]),
    ("tbnn_adapter_create", C.c_int, [C.c_float, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_double, C.c_float, C.c_float, C.c_int32,
                                      C.c_uint64, C.POINTER(_H)]),
    ("tbnn_adapter_destroy", C.c_int, [_H]),
    ("tbnn_adapter_update", C.c_int, [_H, _fp, C.c_int32, C.c_float, C.c_int32, C.c_int32, _fp,
                                      C.POINTER(C.c_int32), _fp]),
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m tensorbnn_amd.build` "
            "(hipcc --offload-arch=gfx950).  tensorbnn_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    # the ABI version first: a stale library (built before the header grew) must say so, not die on a missing symbol
    try:
        lib.tbnn_abi_version.restype = C.c_int
        have = lib.tbnn_abi_version()
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} implements C-ABI version {have}, this binding needs {ABI_VERSION} (include/tbnn.h): "
                          "rebuild it with `python -m tensorbnn_amd.build --force`")
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


_QUOTA_WARNED = False


def cpu_quota():
    """CPUs this container may use per scheduling period (cgroup v2 cpu.max), or None when unlimited / unknown"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError, IndexError):
        return None


def _warn_cpu_quota_once():
    """A chain is driven by ONE host thread that queues a few hundred launches per transition.  Under a CPU quota a math library with a
    thread per logical CPU (NumPy's OpenBLAS on a 256-CPU box with a 16-CPU quota) exhausts the quota period in milliseconds and the kernel
    freezes the whole container -- that thread too -- until the period ends: single 80-ms stalls (NOTES.md round 4).  Said
    once per process when a BLAS / OpenMP pool larger than the quota is loaded."""
    global _QUOTA_WARNED
    if _QUOTA_WARNED:
        return
    _QUOTA_WARNED = True
    quota = cpu_quota()
    if quota is None or (os.cpu_count() or 1) <= quota:
        return
    try:
        from threadpoolctl import threadpool_info
        big = [f"{i.get('internal_api')}: {i.get('num_threads')} threads" for i in threadpool_info() if (i.get("num_threads") or 0) > quota]
    except Exception:
        return
    if big:
        import warnings
        warnings.warn(f"tensorbnn_amd: this container may use {quota:g} CPUs per period but a math library runs {', '.join(big)}; its threads can "
                      f"get the whole container throttled, the GPU launch thread included (single ~80 ms stalls).  Set OPENBLAS_NUM_THREADS / "
                      f"OMP_NUM_THREADS / MKL_NUM_THREADS <= {int(quota)} before importing NumPy (INTEGRATION.md section 4a).", RuntimeWarning, stacklevel=3)


def _check(rc: int):
    if rc < 0:
        raise TbnnError(f"libtbnn error {rc}: {lib.tbnn_last_error().decode()}")
    return rc


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_fp)


def build_id() -> str:
    """hash of the sources the loaded libtbnn.so was built from (tbnn_build_id)"""
    return lib.tbnn_build_id().decode()


def lint_status() -> str:
    """what the build-time MFMA hazard check did to the kernel units of the loaded library (tbnn_lint_status)"""
    return lib.tbnn_lint_status().decode()


def device_count() -> int:
    rc = lib.tbnn_device_count()
    return max(rc, 0)


class Chain:
    """One HMC chain on one device (tbnn_handle)."""

    def __init__(self, layers: Sequence[tuple], likelihood: int = LIK_GAUSSIAN, fixed_sd: float = 0.1,
                 device: int = 0, seed: int = 50, chain_id: int = 0, kernel: int = KERNEL_AUTO,
                 jit: Optional[bool] = None):
        """jit: compile + register MFMA kernels for a shape outside the ahead-of-time registries (jit.py);
        None -> the TBNN_JIT environment switch (default on); only consulted for KERNEL_AUTO / KERNEL_FAST."""
        arr = (LayerDesc * len(layers))(*[LayerDesc(*map(int, l)) for l in layers])
        self._layers_keepalive = arr
        desc = NetDesc(len(layers), arr, int(likelihood), float(fixed_sd), int(kernel), 0)
        if kernel != KERNEL_GENERIC:
            from . import jit as _jit
            if (jit if jit is not None else _jit.enabled()) and lib.tbnn_fused_kernel_available(C.byref(desc)) == 0:
                _jit.ensure_registered(layers, likelihood)
        _warn_cpu_quota_once()
        h = _H()
        _check(lib.tbnn_create(C.byref(desc), int(device), int(seed), int(chain_id), C.byref(h)))
        self._h = h
        self.P = lib.tbnn_param_count(h)
        self.H = lib.tbnn_hyper_count(h)
        self.d_in = int(layers[0][0])
        self.d_out = int(layers[-1][1])
        self.n = 0
        if kernel == KERNEL_AUTO and self.kernel_name == "generic" and len(layers) >= 2:
            # loud, not silent: the generic thread-per-row kernel is ~100x off the MFMA kernels; only
            # reached with TBNN_LAYERED=0 -- otherwise tbnn_create picks the layered MFMA family for such shapes
            import warnings
            warnings.warn(f"tensorbnn_amd: network {[int(layers[0][0])] + [int(l[1]) for l in layers]} runs on the generic "
                          "thread-per-row kernel: no MFMA kernel family covers this shape, or hipcc is not available to "
                          "instantiate one (tensorbnn_amd/jit.py)", RuntimeWarning, stacklevel=2)

    @property
    def kernel_name(self) -> str:
        return lib.tbnn_kernel_name(self._h).decode()

    @property
    def last_transition_path(self) -> str:
        """"per-step" | "trajectory": the kernels that ran the last transition's leapfrog steps"""
        return lib.tbnn_last_transition_path(self._h).decode()

    def close(self):
        if getattr(self, "_h", None):
            lib.tbnn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data / state
    def set_data(self, X, Y):
        X = _f32(X).reshape(-1, self.d_in)
        Y = _f32(Y).reshape(X.shape[0], self.d_out)
        self.n = X.shape[0]
        _check(lib.tbnn_set_data(self._h, _p(X), _p(Y), self.n))

    def set_data_device(self, dX_ptr: int, dY_ptr: int, n: int):
        self.n = int(n)
        _check(lib.tbnn_set_data_device(self._h, C.c_void_p(dX_ptr), C.c_void_p(dY_ptr), self.n))

    def set_state(self, theta):
        theta = _f32(theta).reshape(-1)
        assert theta.size == self.P, (theta.size, self.P)
        _check(lib.tbnn_set_state(self._h, _p(theta)))

    def get_state(self) -> np.ndarray:
        out = np.empty(self.P, dtype=np.float32)
        _check(lib.tbnn_get_state(self._h, _p(out)))
        return out

    def fused_burst_us(self, reps: int = 100) -> float:
        """microseconds per fused forward+backward pass at the current state: `reps` passes back to back between one pair of events"""
        us = C.c_float(0.0)
        _check(lib.tbnn_debug_fused_burst(self._h, int(reps), C.byref(us)))
        return float(us.value)

    def debug_momentum(self) -> np.ndarray:
        """the momentum the last hmc_step's trajectory ended with (tests: time reversal)"""
        out = np.empty(self.P, dtype=np.float32)
        _check(lib.tbnn_debug_momentum(self._h, _p(out)))
        return out

    def set_hypers(self, eta):
        eta = _f32(eta).reshape(-1)
        assert eta.size == self.H, (eta.size, self.H)
        _check(lib.tbnn_set_hypers(self._h, _p(eta)))

    def get_hypers(self) -> np.ndarray:
        out = np.empty(self.H, dtype=np.float32)
        _check(lib.tbnn_get_hypers(self._h, _p(out)))
        return out

    # ---- evaluation
    def logp_grad(self, theta=None, eta=None):
        th = None if theta is None else _f32(theta).reshape(-1)
        et = None if eta is None else _f32(eta).reshape(-1)
        lp, st = C.c_double(), C.c_double()
        g = np.empty(self.P, dtype=np.float32)
        _check(lib.tbnn_logp_grad(self._h, _p(th), _p(et), C.byref(lp), _p(g), C.byref(st)))
        return lp.value, g, st.value

    def forward(self, X, theta=None) -> np.ndarray:
        X = _f32(X).reshape(-1, self.d_in)
        th = None if theta is None else _f32(theta).reshape(-1)
        out = np.empty((self.d_out, X.shape[0]), dtype=np.float32)
        _check(lib.tbnn_forward(self._h, _p(th), _p(X), X.shape[0], _p(out)))
        return out

    def hyper_logp_grad(self, eta=None):
        et = None if eta is None else _f32(eta).reshape(-1)
        lp = C.c_double()
        g = np.empty(self.H, dtype=np.float32)
        _check(lib.tbnn_hyper_logp_grad(self._h, _p(et), C.byref(lp), _p(g)))
        return lp.value, g

    # ---- transitions
    def hmc_step(self, eps: float, L: int, p0=None, log_u=None, trace: bool = False):
        p0a = None if p0 is None else _f32(p0).reshape(-1)
        lua = None if log_u is None else _f32([log_u])
        out = StepOut()
        tr = np.empty(L + 1, dtype=np.float64) if trace else None
        _check(lib.tbnn_hmc_step(self._h, float(eps), int(L), _p(p0a), _p(lua), C.byref(out),
                                 None if tr is None else tr.ctypes.data_as(_dp)))
        d = out.as_dict()
        if trace:
            d["trace_logp"] = tr
        return d

    def hmc_run(self, eps: float, L: int, n_epochs: int):
        outs = (StepOut * n_epochs)()
        _check(lib.tbnn_hmc_run(self._h, float(eps), int(L), int(n_epochs), outs))
        return [o.as_dict() for o in outs]

    def hyper_step(self, eps_h: float, L_h: int, p0=None, log_u=None):
        p0a = None if p0 is None else _f32(p0).reshape(-1)
        lua = None if log_u is None else _f32([log_u])
        out = StepOut()
        _check(lib.tbnn_hyper_step(self._h, float(eps_h), int(L_h), _p(p0a), _p(lua), C.byref(out)))
        return out.as_dict()

    def export_sample_device(self, d_ptr: int):
        _check(lib.tbnn_export_sample_device(self._h, C.c_void_p(d_ptr)))

    def debug_draw(self, epoch: int, purpose: int, n: int):
        out = np.empty(n, dtype=np.float32)
        lu = C.c_float()
        _check(lib.tbnn_debug_draw(self._h, epoch, purpose, n, _p(out), C.byref(lu)))
        return out, lu.value

    def set_epoch(self, epoch: int):
        _check(lib.tbnn_set_epoch(self._h, int(epoch)))

    def set_profiling(self, stride: int):
        _check(lib.tbnn_set_profiling(self._h, int(stride)))

    def set_validation(self, X, Y):
        X = _f32(X).reshape(-1, self.d_in)
        Y = _f32(Y).reshape(X.shape[0], self.d_out)
        _check(lib.tbnn_set_validation(self._h, _p(X), _p(Y), X.shape[0]))
        self.nv = X.shape[0]

    def predict(self, which: int = 0, theta=None) -> np.ndarray:
        """[d_out, n] predictions over the staged training (0) / validation (1) rows"""
        n = self.nv if which else self.n
        out = np.empty((self.d_out, n), dtype=np.float32)
        th = None if theta is None else _f32(theta).reshape(-1)
        _check(lib.tbnn_predict(self._h, int(which), _p(th), _p(out)))
        return out

    def forward_many(self, thetas, X=None, which: int = 1) -> np.ndarray:
        """predictions of an ensemble: thetas [m, P] -> [m, d_out, rows]; X None: the staged rows (0 train, 1 validation)"""
        th = np.ascontiguousarray(thetas, dtype=np.float32)
        if th.ndim != 2 or th.shape[1] != self.P:
            raise ValueError(f"thetas must be [m, {self.P}]")
        if X is None:
            n, xp = (self.nv if which else self.n), None
        else:
            xp = _f32(X).reshape(-1, self.d_in)
            n = xp.shape[0]
        out = np.empty((th.shape[0], self.d_out, n), dtype=np.float32)
        _check(lib.tbnn_forward_many(self._h, _p(th), th.shape[0], th.shape[1], int(which), _p(xp), n, _p(out)))
        return out

    def hyper_probs_many(self, thetas, etas, priors=None) -> np.ndarray:
        """sum over the dense layers of calculateHyperProbs for m saved networks (predictor.trainProbs / reweight):
        thetas [m, P], etas [m, >= 4 * layers], priors: one PRIOR_* per dense layer or None (the chain's) -> float64 [m]"""
        th = np.ascontiguousarray(thetas, dtype=np.float32)
        et = np.ascontiguousarray(etas, dtype=np.float32)
        if th.ndim != 2 or th.shape[1] != self.P or et.ndim != 2 or et.shape[0] != th.shape[0]:
            raise ValueError(f"thetas must be [m, {self.P}] and etas [m, >= 4 * layers]")
        pr = None if priors is None else (C.c_int32 * len(priors))(*[int(x) for x in priors])
        out = np.empty(th.shape[0], dtype=np.float64)
        _check(lib.tbnn_hyper_probs_many(self._h, pr, _p(th), th.shape[1], _p(et), et.shape[1], th.shape[0],
                                         out.ctypes.data_as(_dp)))
        return out

    def metrics(self, which: int = 0, theta=None, mean: float = 0.0, sd: float = 1.0, exp_pred: bool = False,
                exp_real: bool = False):
        """(mean squared error, mean percent error, mean |r - round(p)|) over the staged rows (metrics.py:30-141)"""
        out = (C.c_double * 3)()
        th = None if theta is None else _f32(theta).reshape(-1)
        _check(lib.tbnn_metrics(self._h, int(which), _p(th), float(mean), float(sd), int(exp_pred), int(exp_real), out))
        return float(out[0]), float(out[1]), float(out[2])

    def gather_samples(self, comm: "Comm", d_out_ptr: int = 0) -> np.ndarray:
        """RCCL all-gather of (theta, eta) over the communicator: returns [world, P+H]"""
        out = np.empty((comm.world, self.P + self.H), dtype=np.float32)
        _check(lib.tbnn_gather_samples(self._h, comm._c, C.c_void_p(d_out_ptr) if d_out_ptr else None, _p(out)))
        return out

    def set_row_shard(self, comm: Optional["Comm"], n_total: int = 0):
        """row-sharded single chain: this rank's set_data rows are one block of n_total rows (None: back to unsharded)"""
        _check(lib.tbnn_set_row_shard(self._h, comm._c if comm is not None else None, int(n_total)))


class ChainGroup:
    """n_chains independent chains of one network on ONE device behind one handle (tbnn_create_multi): chain c is bit for bit
    what Chain(..., chain_id=chain_id + c) would be; their per-chain kernels run as one launch each (gridDim.y = chain), so
    small problems -- bound by launch latency, most of the GPU idle -- advance n_chains chains in the time of one."""

    def __init__(self, layers: Sequence[tuple], n_chains: int, likelihood: int = LIK_GAUSSIAN, fixed_sd: float = 0.1,
                 device: int = 0, seed: int = 50, chain_id: int = 0, kernel: int = KERNEL_AUTO, jit: Optional[bool] = None):
        arr = (LayerDesc * len(layers))(*[LayerDesc(*map(int, l)) for l in layers])
        self._layers_keepalive = arr
        desc = NetDesc(len(layers), arr, int(likelihood), float(fixed_sd), int(kernel), 0)
        if kernel != KERNEL_GENERIC:
            from . import jit as _jit
            if (jit if jit is not None else _jit.enabled()) and lib.tbnn_fused_kernel_available(C.byref(desc)) == 0:
                _jit.ensure_registered(layers, likelihood)
        _warn_cpu_quota_once()
        h = _H()
        _check(lib.tbnn_create_multi(C.byref(desc), int(device), int(seed), int(chain_id), int(n_chains), C.byref(h)))
        self._h = h
        self.C = lib.tbnn_chain_count(h)
        self.P = lib.tbnn_param_count(h)
        self.H = lib.tbnn_hyper_count(h)
        self.d_in, self.d_out, self.n = int(layers[0][0]), int(layers[-1][1]), 0

    @property
    def kernel_name(self) -> str:
        return lib.tbnn_kernel_name(self._h).decode()

    @property
    def last_transition_path(self) -> str:
        """"per-step" | "trajectory": the kernels that ran the last transition's leapfrog steps"""
        return lib.tbnn_last_transition_path(self._h).decode()

    def close(self):
        if getattr(self, "_h", None):
            lib.tbnn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_data(self, X, Y):
        X = _f32(X).reshape(-1, self.d_in)
        Y = _f32(Y).reshape(X.shape[0], self.d_out)
        self.n = X.shape[0]
        _check(lib.tbnn_set_data(self._h, _p(X), _p(Y), self.n))

    def set_data_device(self, dX_ptr: int, dY_ptr: int, n: int):
        self.n = int(n)
        _check(lib.tbnn_set_data_device(self._h, C.c_void_p(dX_ptr), C.c_void_p(dY_ptr), self.n))

    def set_state(self, thetas):
        """[n_chains, P], or [P] for every chain alike"""
        th = np.ascontiguousarray(np.broadcast_to(_f32(thetas).reshape(-1, self.P), (self.C, self.P)))
        _check(lib.tbnn_set_state(self._h, _p(th)))

    def get_state(self) -> np.ndarray:
        out = np.empty((self.C, self.P), dtype=np.float32)
        _check(lib.tbnn_get_state(self._h, _p(out)))
        return out

    def set_hypers(self, etas):
        et = np.ascontiguousarray(np.broadcast_to(_f32(etas).reshape(-1, self.H), (self.C, self.H)))
        _check(lib.tbnn_set_hypers(self._h, _p(et)))

    def get_hypers(self) -> np.ndarray:
        out = np.empty((self.C, self.H), dtype=np.float32)
        _check(lib.tbnn_get_hypers(self._h, _p(out)))
        return out

    def hmc_run(self, eps: float, L: int, n_epochs: int):
        """n_epochs transitions of every chain with no host round trip: list over chains of lists of records"""
        outs = (StepOut * (self.C * n_epochs))()
        _check(lib.tbnn_hmc_run(self._h, float(eps), int(L), int(n_epochs), outs))
        return [[outs[c * n_epochs + e].as_dict() for e in range(n_epochs)] for c in range(self.C)]

    def hmc_step(self, eps: float, L: int):
        outs = (StepOut * self.C)()
        _check(lib.tbnn_hmc_step(self._h, float(eps), int(L), None, None, outs, None))
        return [o.as_dict() for o in outs]

    def hyper_step(self, eps_h: float, L_h: int):
        outs = (StepOut * self.C)()
        _check(lib.tbnn_hyper_step(self._h, float(eps_h), int(L_h), None, None, outs))
        return [o.as_dict() for o in outs]

    def set_epoch(self, epoch: int):
        _check(lib.tbnn_set_epoch(self._h, int(epoch)))

    # ---- every chain at its own (eps, L) / its own hyper step size: one adapter and one dual averaging per chain, as C runs of
    # the reference would have (network.py:221-235, :603-607, :457-469)
    def _each(self, eps, L):
        e = np.ascontiguousarray(np.broadcast_to(np.asarray(eps, dtype=np.float32), (self.C,)))
        l = np.ascontiguousarray(np.broadcast_to(np.asarray(L, dtype=np.int32), (self.C,)))
        return e, l

    def hmc_step_each(self, eps, L):
        """eps, L: one value per chain; chain c == the solo chain chain_id + c driven with (eps[c], L[c])"""
        e, l = self._each(eps, L)
        outs = (StepOut * self.C)()
        _check(lib.tbnn_hmc_step_each(self._h, _p(e), l.ctypes.data_as(C.POINTER(C.c_int32)), outs))
        return [o.as_dict() for o in outs]

    def hmc_run_each(self, eps, L, n_epochs: int):
        e, l = self._each(eps, L)
        outs = (StepOut * (self.C * n_epochs))()
        _check(lib.tbnn_hmc_run_each(self._h, _p(e), l.ctypes.data_as(C.POINTER(C.c_int32)), int(n_epochs), outs))
        return [[outs[c * n_epochs + k].as_dict() for k in range(n_epochs)] for c in range(self.C)]

    def hyper_step_each(self, eps_h, L_h: int):
        e = np.ascontiguousarray(np.broadcast_to(np.asarray(eps_h, dtype=np.float32), (self.C,)))
        outs = (StepOut * self.C)()
        _check(lib.tbnn_hyper_step_each(self._h, _p(e), int(L_h), outs))
        return [o.as_dict() for o in outs]


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _check(lib.tbnn_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """tbnn_comm_* : an RCCL communicator bound to a chain's device and stream (librccl.so is dlopen'ed on first use)"""

    def __init__(self, chain: "Chain", world: int, rank: int, uid: bytes):
        if len(uid) != COMM_ID_BYTES:
            raise ValueError("uid must be COMM_ID_BYTES long")
        self.world, self.rank = int(world), int(rank)
        self._c = _H()
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(uid)
        _check(lib.tbnn_comm_create(chain._h, self.world, self.rank, buf, C.byref(self._c)))

    def count(self) -> int:
        """ranks in the communicator as the collective library reports them (ncclCommCount)"""
        return _check(lib.tbnn_comm_count(self._c))

    def close(self):
        if getattr(self, "_c", None):
            lib.tbnn_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Adapter:
    """tbnn_adapter_* : the (eps, L) GP-UCB adapter (paramAdapter.py:11-292) in host C++."""

    def __init__(self, e1, L1, el, eu, eNumber, Ll, Lu, lStep, m, k, a=4, delta=0.1, randomSteps=10, seed=0):
        h = _H()
        _check(lib.tbnn_adapter_create(float(e1), int(L1), float(el), float(eu), int(eNumber), int(Ll), int(Lu),
                                       int(lStep), int(m), float(k), float(a), float(delta), int(randomSteps),
                                       int(seed), C.byref(h)))
        self._h = h

    def update(self, state, inject_u: float = -1.0, inject_e: int = -1, inject_l: int = -1):
        s = _f32(state).reshape(-1)
        e, L, sjd = C.c_float(), C.c_int32(), C.c_float()
        _check(lib.tbnn_adapter_update(self._h, _p(s), s.size, float(inject_u), int(inject_e), int(inject_l),
                                       C.byref(e), C.byref(L), C.byref(sjd)))
        return e.value, L.value, sjd.value

    def close(self):
        if getattr(self, "_h", None):
            lib.tbnn_adapter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
