"""Run-time instantiation of the shape-specialised MFMA kernels (`python -m tensorbnn_amd.jit 5,50,50,1`).

The fused forward+backward kernels are C++ templates over the network shape (csrc/kernels_fast*.hpp,
csrc/kernels_wide.hpp); libtbnn.so carries ahead-of-time instantiations for BASELINE's configs only.
For any other network this module writes a ten-line translation unit, compiles it with hipcc for
gfx950 into a cached shared object and registers it with the library (tbnn_register_kernel_lib), so
that `Chain(..., kernel=KERNEL_AUTO)` runs on a FUSED kernel instead of the layered run-time-shape MFMA kernels
(csrc/kernels_layered.hpp: what any architecture without a fused kernel runs on, 2-3x slower where both apply).
(The reference gets there through tf.function tracing + XLA, network.py:359-362; here the kernels
are hand-written and only their *shape parameters* are bound at run time.)

Family choice (first that compiles wins, in the order below -- except that a narrow network with 3 .. 16 outputs tries mid first: `families`; a shape
no family accepts is remembered as `.fail` and runs on the layered kernels):
  * narrow (`k_fwd_bwd_fast3`, else `k_fwd_bwd_fast`): every dW accumulator in one wave's registers --
    fan-in <= 16, widths <= 64 (a network with ONE hidden layer: <= 256 units, <= 160 with more than two outputs), at most NARROW_TILES 16x16 dW tiles in total;
  * mid (`k_fwd_bwd_mid`): >= 3 dense layers, <= 16 outputs (3 .. 16: the last layer is an MFMA layer too), fan-in <= 128, at most 63 dW tiles over the MFMA
    layers and weight images + operand blocks within 160 KB of LDS (`mid_fits`): one fused kernel, nothing through HBM;
  * tall (`k_fwd_bwd_tall`): a first-layer fan-in above the narrow family's 16 (.. a few thousand columns) in front of narrow hidden layers
    (<= 64 units), <= 16 outputs (3 .. 16: the last layer is an MFMA layer too): the fan-in split over the four waves of a workgroup, W_0 and dW_0 in registers
    (`tall_fits`) -- the reference's MNIST example 784 -> 20 -> 20 -> 1;
  * wide (`k_chain_wide` + `k_dw_wide`): >= 3 dense layers, <= 16 outputs (3 .. 16: the last layer is one more streamed middle layer), fan-in <= 32,
    hidden widths <= 256.
Requirements common to all: dense layers only; hidden layers that do not all carry the same activation get a packed per-layer code (`shape_of`), nine
hidden layers at most.
"""
import hashlib
import os
import re
import subprocess
import sys
from typing import Optional, Sequence

NARROW_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form"]      # keep in step with build.py
TALL_NOP = ["-DTBNN_ASM_MFMA_NOP=0"]                      # the tall family: no wait states inside its asm MFMAs (build.py says why); the check repairs

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
NARROW_TILES = 40
MID_MAX_FANIN = 128          # keep in step with kernels_mid.hpp


def cache_dir() -> str:
    d = os.environ.get("TBNN_JIT_DIR", os.path.join(HERE, "_jit"))
    os.makedirs(d, exist_ok=True)
    return d


def enabled() -> bool:
    return os.environ.get("TBNN_JIT", "1") != "0"


def _cdiv(a, b):
    return (a + b - 1) // b


def _sources_stamp() -> str:
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hpp"):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "tbnn.h"), "rb").read())
    for f in ("hazard_lint.py", "checked_compile.py"):          # the check is part of the compile: new rules, new libraries
        h.update(open(os.path.join(HERE, f), "rb").read())
    return h.hexdigest()


ACT_PACKED = 0x40000000          # csrc/common.hpp: TBNN_ACT_PACKED


def shape_of(layers: Sequence[tuple], likelihood: int):
    """(dims, hact, lact, bern) or None when the fused kernels cannot express the network; hact: the hidden layers' activation, or the
    packed per-layer code when they differ"""
    from . import _native as nat
    dims = [int(layers[0][0])] + [int(l[1]) for l in layers]
    acts = [int(l[2]) for l in layers]
    if len(layers) < 2:
        return None
    hact = acts[0]
    if any(a != hact for a in acts[:-1]):
        # hidden layers with different activations (network.add takes any sequence: tensorBNN/network.py:173-191): the packed per-layer
        # code of csrc/kernels_fast.hpp (Shape::act), 3 bits per hidden layer, 9 layers at most
        if len(acts) - 1 > 9 or any(not 0 <= a <= 7 for a in acts[:-1]):
            return None
        hact = ACT_PACKED | sum(a << (3 * l) for l, a in enumerate(acts[:-1]))
    return dims, hact, acts[-1], int(likelihood == nat.LIK_BERNOULLI)


def families(dims) -> list:
    """candidate kernel families for `dims`, best first"""
    nl = len(dims) - 1
    out = []
    tiles = sum(_cdiv(dims[l + 1], 16) * _cdiv(dims[l] + 1, 16) for l in range(nl))
    # (widths: 64, the widest layer the hand-threaded dW phases of deeper networks were written and fuzzed for; a network with ONE hidden layer has
    # none of those phases between two wide layers and instantiates well beyond that -- 1 -> 100 -> 1, the canonical BNN regression demo, ran on the
    # layered family until late round 6: 73 us per step at 1e5 rows against 17 for 1 -> 64 -> 1)
    # (one hidden layer: 256 units with one or two outputs, 160 with 3 .. 16 -- beyond those a random draw of 60 such shapes had 9 refused by the build:
    # fast3 spills from ~300 units, k_fwd_bwd_fast's LDS plan overflows with many outputs behind > 200 units)
    narrow = dims[0] <= 16 and tiles <= NARROW_TILES and (max(dims) <= 64 or (nl == 2 and max(dims) <= (256 if dims[-1] <= 2 else 160)))
    mid = nl >= 3 and dims[-1] <= 16 and dims[0] <= MID_MAX_FANIN and mid_fits(dims)
    # 3 .. 16 outputs on a narrow network: fast3 does not take them, and the mid-width kernel (MFMA last layer, round 6) measures 7 - 13 % ahead of
    # k_fwd_bwd_fast there (5 -> 50 -> 50 -> 50 -> 3 at 1e5 rows 65.3 against 70.4 us per step, 8 -> 40 -> 40 -> 10 34.8 against 39.8): mid first
    mid_first = narrow and mid and dims[-1] > 2
    if mid_first:
        out.append("mid")
    if narrow:
        if dims[-1] <= 2 and nl >= 2:
            out.append("fast3")
        out.append("fast")
    if mid and not mid_first:
        out.append("mid")
    # (fan-in 17 .. 32 with ONE hidden layer: the narrow family stops at 16 inputs, mid and wide need two hidden layers -- the tall kernel is what such a
    # network has; late round 6, measured against the layered family.  Deeper networks of that fan-in keep their round-4 order: mid, then wide)
    if nl >= 2 and dims[-1] <= 16 and dims[0] > (16 if nl == 2 else 32) and tall_fits(dims):
        out.append("tall")
    # (3 .. 16 outputs: the last layer as one more middle layer, round 6.  Fan-in: 32 was round 1's choice -- x in registers, W_0 in LDS, dW_0 in AccVGPRs --;
    # the kernels build and hold for fan-in up to 128 wherever W_0 fits the LDS next to the ring: late round 6, `wide_fits`)
    if nl >= 3 and dims[-1] <= 16 and dims[0] <= 128 and max(dims[1:-1]) <= 256 and wide_fits(dims):
        out.append("wide")
    skip = {f for f in os.environ.get("TBNN_JIT_SKIP", "").split(",") if f}      # diagnostic / tests: e.g. "mid" forces the wide path
    return [f for f in out if f not in skip]


def wide_fits(dims) -> bool:
    """fan-in above 32 on the wide family (kernels_wide.hpp, WideCfg): W_0's operand granules (ceil(out_0 / 16) x ceil(d_in / 16) KB) in LDS next to the
    weight ring (4 slots of the widest layer's granules) and the per-wave scratch, and dW_0's ceil(out_0 / 16) x ceil((d_in + 1) / 16) accumulator
    tiles beside the chain's registers (an estimate: the build refuses what does not fit or spills, and the next family takes the shape)"""
    if dims[0] <= 32:
        return True
    mt0, kg0, nt0 = _cdiv(dims[1], 16), _cdiv(dims[0], 16), _cdiv(dims[0] + 1, 16)
    maxgran = 4 * _cdiv(max(_cdiv(d, 16) for d in dims[1:-1]), 4)
    lds = (mt0 * kg0 * 256 + 4 * maxgran * 256 + 4 * (16 * (16 * nt0 + 4) + 16 * 68) + 16 * sum(_cdiv(d, 16) for d in dims[1:])) * 4
    return lds <= 156 * 1024 and mt0 * nt0 <= 63


def mid_fits(dims) -> bool:
    """the mid-width fused kernel (kernels_mid.hpp, MidCfg): every dW tile of the MFMA layers in one wave's AccVGPRs (<= 63
    tiles) and the weight images + per-wave operand blocks in 160 KB of LDS.  <= 2 outputs: the last layer runs on the VALU; 3 .. 16
    outputs: it is one more MFMA layer (one output tile)"""
    nl = len(dims) - 1
    vl = dims[-1] <= 2
    nm = nl - 2 if vl else nl - 1               # MFMA layers behind layer 0: 1 .. nm
    tr = lambda l: _cdiv(dims[l], 16)           # tiles of a_l (input of layer l)
    ta = lambda l: _cdiv(dims[l] + 1, 16)
    tiles = tr(1) * ta(0) + sum(tr(l + 1) * ta(l) for l in range(1, nm + 1))
    if tiles > 63:
        return False
    r4 = lambda a: (a + 3) & ~3
    perm = r4(tr(1) * _cdiv(dims[0], 16) * 256 + sum(16 * tr(l + 1) for l in range(nm + 1)) + (dims[-1] * 16 * tr(nl - 1) + dims[-1] if vl else 0))
    img = r4(perm + sum(16 * tr(l + 1) * (16 * tr(l) + 4) for l in range(1, nm + 1)))
    maxt = max(tr(l) for l in range(1, nl if vl else nl + 1))
    wave = (ta(0) + sum(ta(l) for l in range(1, nm + 1)) + 2 * maxt) * 256      # (MidCfg::WAVE_FLOATS: two delta regions)
    return (img + 4 * wave) * 4 + 64 <= 160 * 1024


def tall_fits(dims) -> bool:
    """the tall-fan-in fused kernel (kernels_tall.hpp, TallCfg): a wave's chunk of W_0 and of dW_0 in its registers, the narrow
    layers' images, the exchange buffers and the per-wave operand blocks in 160 KB of LDS (an estimate: the build refuses a
    kernel that spills and the next family takes the shape).  <= 2 outputs: the last layer on the VALU; 3 .. 16: one more MFMA layer"""
    nl = len(dims) - 1
    if max(dims[1:-1]) > (128 if nl == 2 else 64):      # (one hidden layer: up to 128 units where the estimates below hold -- late round 6, as `families`)
        return False
    vl = dims[-1] <= 2
    nm = nl - 2 if vl else nl - 1               # MFMA layers behind layer 0: 1 .. nm
    tr = lambda l: _cdiv(dims[l], 16)
    ta = lambda l: _cdiv(dims[l] + 1, 16)
    mt0, ch = tr(1), _cdiv(ta(0), 4)
    dwm = sum(tr(l + 1) * ta(l) for l in range(1, nm + 1))
    vgpr = 4 * mt0 * ch + 6 * ch + 70
    agpr = 4 * mt0 * ch + 4 * dwm
    if vgpr > 256 or vgpr + agpr > 500:
        return False
    r4 = lambda a: (a + 3) & ~3
    perm = r4(sum(16 * tr(l + 1) for l in range(nm + 1)) + (dims[-1] * 16 * tr(nl - 1) + dims[-1] if vl else 0))
    small = r4(perm + sum(16 * tr(l + 1) * (16 * tr(l) + 4) for l in range(1, nm + 1)))
    # exchange buffer of a group's 4 tiles | dW_0 staging of the epilogue, delta_0 of the group, per-wave a_l / delta_l blocks of the middle layers
    wave = (sum(ta(l) for l in range(1, nm + 1)) + sum(tr(l + 1) for l in range(1, nm + 1))) * 256
    return (small + max(16 * mt0, 4 * ch) * 256 + 4 * mt0 * 256 + 4 * wave) * 4 + 64 <= 160 * 1024


def source(dims, hact, lact, bern, family) -> str:
    shape = f"Shape<{hact}, {lact}, {'true' if bern else 'false'}, {', '.join(map(str, dims))}>"
    if family == "wide":
        return (f'#include "{CSRC}/jit_wide.hpp"\nusing S = {shape};\n'
                'extern "C" int tbnn_jit_ops(FusedOps* o) { JitWide<S>::fill(o); return 0; }\n')
    if family == "tall":
        return (f'#include "{CSRC}/jit_tall.hpp"\nusing S = {shape};\n'
                'extern "C" int tbnn_jit_ops(FusedOps* o) { JitTall<S>::fill(o); return 0; }\n')
    if family == "mid":
        return (f'#include "{CSRC}/jit_mid.hpp"\nusing S = {shape};\n'
                'extern "C" int tbnn_jit_ops(FusedOps* o) { JitMid<S>::fill(o); return 0; }\n')
    f3 = "true" if family == "fast3" else "false"
    return (f'#include "{CSRC}/jit_narrow.hpp"\nusing S = {shape};\n'
            f'extern "C" int tbnn_jit_ops(FusedOps* o) {{ JitNarrow<S, {f3}>::fill(o); return 0; }}\n')


_WARNED = set()


def _warn_generic(dims, why):
    """once per shape and process, as a RuntimeWarning (visible to `-W error`, logging.captureWarnings, pytest): the layered
    family is 2-3x slower than a fused kernel where both apply, and only the kernel's name would tell otherwise"""
    import warnings
    key = (tuple(dims), why.split(":")[0])
    if key in _WARNED:
        return
    _WARNED.add(key)
    warnings.warn(f"tensorbnn_amd: network {list(dims)} runs on the layered run-time-shape MFMA kernels (kernels_layered.hpp), "
                  f"not on a fused kernel: {why}", RuntimeWarning, stacklevel=3)


def build(layers: Sequence[tuple], likelihood: int, verbose: bool = False) -> Optional[str]:
    """path of the compiled kernel library for this network, or None (layered kernels, with a note on stderr)"""
    import fcntl
    sh = shape_of(layers, likelihood)
    if sh is None:
        _warn_generic([int(layers[0][0])] + [int(l[1]) for l in layers],
                      "the fused kernels need >= 2 dense layers (and at most 9 hidden layers when their activations differ)")
        return None
    dims, hact, lact, bern = sh
    extra = os.environ.get("TBNN_JIT_FLAGS", "").split()          # diagnostic builds (-DTBNN_WPAD=8 ...); part of the cache key
    if os.environ.get("TBNN_JIT_LOG"):                            # which shapes a run asked for (tests/jit_shapes.json is made from this: `prebuild`)
        import json
        with open(os.environ["TBNN_JIT_LOG"], "a") as f:
            f.write(json.dumps({"layers": [list(map(int, l)) for l in layers], "likelihood": int(likelihood),
                                "skip": os.environ.get("TBNN_JIT_SKIP", ""), "flags": os.environ.get("TBNN_JIT_FLAGS", "")}) + "\n")
    key = hashlib.sha1(f"{dims}|{hact}|{lact}|{bern}|{_sources_stamp()}|{extra}|{families(dims)}|{NARROW_FLAGS}|{TALL_NOP}".encode()).hexdigest()[:20]
    d = cache_dir()
    so, failed = os.path.join(d, f"tbnn_{key}.so"), os.path.join(d, f"tbnn_{key}.fail")
    if os.path.exists(so):
        return so
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        _warn_generic(dims, f"no ahead-of-time kernel covers this shape and {hipcc} is not present to instantiate one")
        return None                      # no compiler on this machine: not remembered as a failure
    # one builder per shape: the ranks of a node all reach this point together (Chain.__init__ on every rank); the
    # first takes the lock and compiles, the others wait and pick the finished library up
    with open(os.path.join(d, f"tbnn_{key}.lock"), "w") as lockf:
        fcntl.flock(lockf, fcntl.LOCK_EX)
        try:
            if os.path.exists(so):
                return so
            if os.path.exists(failed):
                _warn_generic(dims, f"no kernel family compiles for this shape (diagnostics: {failed})")
                return None
            print(f"tensorbnn_amd: compiling MFMA kernels for network {dims} (once; cached in {d})", file=sys.stderr, flush=True)
            log, deterministic = [], True
            for fam in families(dims):
                # per-process file names: nothing another process may be reading is ever truncated
                src = os.path.join(d, f"tbnn_{key}_{fam}.{os.getpid()}.hip")
                tmp = so + f".{os.getpid()}.tmp"
                with open(src, "w") as f:
                    f.write(source(dims, hact, lact, bern, fam))
                # compiled through checked_compile.run: hipcc's own steps with the MFMA hazard check (hazard_lint.py) between the device
                # listing and the assembler -- wait states inserted where a pair lacks them, the finished library disassembled and checked
                # again; a library is never handed out unchecked
                from . import checked_compile
                notraj = []                                # set when only the optional trajectory kernel of the shape needs scratch memory
                status = ""
                while True:
                    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"]
                    cmd += NARROW_FLAGS + (TALL_NOP if fam == "tall" else []) + extra + notraj      # as build.py compiles the kernels (VGPR-form chain MFMAs)
                    cmd += ["-Rpass-analysis=kernel-resource-usage", "-o", tmp, src]      # the remarks carry each kernel's ScratchSize
                    if verbose:
                        print(" ".join(cmd), flush=True)
                    r = checked_compile.run(cmd)                    # child processes: never an exec of this one
                    rc, err, status = r.rc, r.stderr, r.status
                    # a fused kernel that needs scratch memory has lost its register plan (accumulators demoted to a stack array
                    # are read back without the wait states an MFMA result needs): refuse it, the next family takes the shape
                    per_fn = re.findall(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", err, re.S)
                    spills = [int(v) for _f, v in per_fn]
                    if rc == 0 and any(v > 0 for v in spills):
                        if not notraj and all(int(v) == 0 or "k_traj_" in f for f, v in per_fn):
                            notraj = ["-DTBNN_TRAJ_WAVES=0"]          # the per-step kernels are fine: the same library without the trajectory kernel
                            continue
                        rc, err = 1, f"{src}:1:1: error: kernel family {fam} spills to scratch for this shape ({max(spills)} bytes per lane)\n"
                    break
                if rc == 0 and verbose:
                    print(f"tensorbnn_amd: {fam} kernels of {dims}: {status}", file=sys.stderr, flush=True)
                if rc == 0:
                    with open(so + ".lint", "w") as f:
                        nop_off = fam == "tall" or "-DTBNN_ASM_MFMA_NOP=0" in extra
                        f.write(f"{fam}: {status}; wait states inside the asm MFMAs: {'off (left to the check)' if nop_off else 'on'}\n")
                for f_ in (src,):
                    if os.path.exists(f_):
                        os.remove(f_)
                if rc == 0:
                    os.replace(tmp, so)
                    return so
                if os.path.exists(tmp):
                    os.remove(tmp)
                # a compiler diagnostic ("error:") with a normal exit status is a property of the shape; anything else
                # (killed, out of disk, exec failure) is transient and must not be remembered
                # -- and so is a driver-level "error:" without a source location (an OOM-killed clang child: "clang: error:
                # unable to execute command: Killed"; a full disk: "error: unable to open output file"): only a located
                # diagnostic (<file>:<line>:<col>: error:) says something about the shape
                located = re.search(r"^[^\s:][^:\n]*:\d+:\d+: (fatal )?error:", err, re.M) is not None
                transient = re.search(r"unable to execute command|No space left|Killed|signal|unable to open output file|Cannot allocate memory", err) is not None
                crashed = "PLEASE submit a bug report" in err          # an internal compiler error on this shape's code: it will crash again
                if (rc < 0 or not located or transient) and not crashed:
                    deterministic = False
                log.append(f"[{fam}] rc={rc}\n{err[-2000:]}")
            if deterministic:
                with open(failed + f".{os.getpid()}", "w") as f:
                    f.write(f"sources {_sources_stamp()}\n" + ("\n".join(log) if log else "no kernel family applies to this shape\n"))
                os.replace(failed + f".{os.getpid()}", failed)
                _warn_generic(dims, f"no kernel family compiles for this shape (diagnostics: {failed})")
            else:
                _warn_generic(dims, "the kernel build failed for a transient reason (not cached; it is retried next time):\n" + "\n".join(log)[-1500:])
            return None
        finally:
            fcntl.flock(lockf, fcntl.LOCK_UN)


def _prebuild_one(job):
    import warnings
    layers, likelihood, skip, flags = job
    os.environ["TBNN_JIT_SKIP"], os.environ["TBNN_JIT_FLAGS"] = skip, flags
    os.environ.pop("TBNN_JIT_LOG", None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return build([tuple(l) for l in layers], likelihood) is not None


def prebuild(jobs, processes: int = 6) -> int:
    """Compile (into the cache: tensorbnn_amd/_jit travels with the tree) the kernel libraries of a list of shapes -- dicts as TBNN_JIT_LOG writes
    them -- side by side; returns how many have a library afterwards.  __graft_entry__.build() calls it with tests/jit_shapes.json, so that a GPU
    test run finds its run-time instantiations built and checked instead of compiling them on the GPU box."""
    import json
    from multiprocessing import get_context
    uniq = {json.dumps(j, sort_keys=True): j for j in jobs}
    todo = [(j["layers"], j["likelihood"], j.get("skip", ""), j.get("flags", "")) for j in uniq.values()]
    keep = {k: os.environ.get(k) for k in ("TBNN_JIT_SKIP", "TBNN_JIT_FLAGS")}
    try:
        # fork, not spawn: a spawned worker re-imports __main__, and a caller that runs from stdin or `-c` has none to import (it hangs);
        # the workers only drive compiler subprocesses and never touch a GPU
        with get_context("fork").Pool(max(1, min(processes, len(todo)))) as pool:
            done = pool.map_async(_prebuild_one, todo, chunksize=1).get(timeout=3600)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return sum(done)


def lint_status(path: str) -> str:
    """what the build-time MFMA hazard check did to a cached kernel library (written next to it by `build`)"""
    try:
        return open(path + ".lint").read().strip()
    except OSError:
        return "unknown (library built before the check was part of the compile)"


def ensure_registered(layers: Sequence[tuple], likelihood: int, verbose: bool = False) -> bool:
    """compile (or fetch from the cache) and register the kernels of this network; False: no fused kernel (layered family)"""
    from . import _native as nat
    path = build(layers, likelihood, verbose)
    if path is None:
        return False
    nat._check(nat.lib.tbnn_register_kernel_lib(path.encode()))
    return True


if __name__ == "__main__":
    from . import _native as nat
    if len(sys.argv) > 2 and sys.argv[1] == "--prebuild":
        # python -m tensorbnn_amd.jit --prebuild shapes.json [processes]: compile the kernel libraries of a list of shapes (the lines a run with
        # TBNN_JIT_LOG=<file> wrote, or a JSON list of them) into the cache -- on a build machine, for a target without a compiler: the cache
        # directory (TBNN_JIT_DIR, default tensorbnn_amd/_jit) travels with the package
        import json
        txt = open(sys.argv[2]).read().strip()
        jobs = json.loads(txt) if txt.startswith("[") else [json.loads(l) for l in txt.splitlines() if l.strip()]
        n = prebuild(jobs, int(sys.argv[3]) if len(sys.argv) > 3 else min(8, os.cpu_count() or 4))
        print(f"{n} of {len(jobs)} kernel libraries present in {cache_dir()}")
        sys.exit(0)
    dims = [int(x) for x in sys.argv[1].split(",")]
    act = {"relu": nat.ACT_RELU, "tanh": nat.ACT_TANH, "sigmoid": nat.ACT_SIGMOID}[sys.argv[2] if len(sys.argv) > 2 else "relu"]
    layers = [(dims[i], dims[i + 1], act if i < len(dims) - 2 else nat.ACT_NONE, nat.PRIOR_CAUCHY) for i in range(len(dims) - 1)]
    print(families(dims), build(layers, nat.LIK_GAUSSIAN, verbose=True))
