"""C-ABI checks that need no GPU: the library loads, exports every symbol the
header declares, refuses to create a chain without a gfx950 device (no CPU
fallback), and its host-side adapter reproduces the oracle's trace."""
import os
import re

import numpy as np
import pytest

import tbnn_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "tbnn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tbnn_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(native):
    import ctypes
    lib = ctypes.CDLL(native.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/tbnn.h but not exported"
    bound = {name for name, _, _ in native.SYMBOLS}
    assert set(syms) == bound, (set(syms) ^ bound)
    assert native.lib.tbnn_abi_version() == native.ABI_VERSION == 3


def test_no_cpu_fallback(native):
    if native.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(native.TbnnError) as e:
        native.Chain([(1, 10, 1, 0), (10, 1, 0, 0)])
    assert "no CPU fallback" in str(e.value) or "hip" in str(e.value).lower()


def test_descriptor_validation_happens_before_device_probe(native):
    with pytest.raises(native.TbnnError):
        native.Chain([(1, 10, 1, 0), (11, 1, 0, 0)])       # dims do not chain
    with pytest.raises(native.TbnnError):
        native.Chain([(1, 10, 9, 0)])                      # unknown activation


def test_adapter_matches_oracle_golden_trace(native):
    """paramAdapter.update (paramAdapter.py:199-292): 60 calls, injected uniforms / grid picks."""
    g = np.load(os.path.join(GOLD, "adapter_trace.npz"))
    c = g["ctor"]
    ad = native.Adapter(c[0], int(c[1]), c[2], c[3], int(c[4]), int(c[5]), int(c[6]), int(c[7]), int(c[8]), c[9],
                        a=c[10], delta=c[11], randomSteps=int(c[12]))
    ora = o.ParamAdapter(c[0], int(c[1]), c[2], c[3], int(c[4]), int(c[5]), int(c[6]), int(c[7]), int(c[8]), c[9],
                         a=c[10], delta=c[11], randomSteps=int(c[12]))
    sj = []
    for t in range(len(g["states"])):
        e, L, sjd = ad.update(g["states"][t], inject_u=float(g["uniforms"][t]), inject_e=int(g["ce"][t]), inject_l=int(g["cl"][t]))
        ora._uniforms, ora._choices = [g["uniforms"][t]], [int(g["ce"][t]), int(g["cl"][t])]
        eo, Lo = ora.update(g["states"][t].copy())
        assert (abs(e - g["outs"][t][0]) < 1e-9 and L == int(g["outs"][t][1])), (t, e, L, g["outs"][t])
        assert abs(eo - e) < 1e-9 and int(Lo) == L
        if t > 0:
            sj.append(sjd)
    np.testing.assert_allclose(sj, g["sjd"], rtol=1e-5)


def test_adapter_random_phase_and_reset(native):
    """strike logic: 50 zero-movement epochs past the random phase halve the grid (paramAdapter.py:208-214)"""
    ad = native.Adapter(1e-3, 10, 1e-4, 1e-2, 10, 5, 20, 1, 2, 200, randomSteps=1)
    ora = o.ParamAdapter(1e-3, 10, 1e-4, 1e-2, 10, 5, 20, 1, 2, 200, randomSteps=1)
    st = np.zeros(7, dtype=np.float32)
    for t in range(70):
        ora._uniforms, ora._choices = [0.0], [t % 10, t % 16]
        eo, Lo = ora.update(st.copy())
        e, L, _ = ad.update(st, inject_u=0.0, inject_e=t % 10, inject_l=t % 16)
        assert abs(e - eo) < 1e-9 and L == int(Lo), t
    assert ora.eu < 1e-2      # the reset happened
