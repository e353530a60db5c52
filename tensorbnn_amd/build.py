"""Build libtbnn.so in-tree with hipcc for gfx950 (`python -m tensorbnn_amd.build`); every kernel unit through checked_compile (MFMA hazard check)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "tbnn_api.hip"), os.path.join(HERE, "csrc", "tbnn_wide.hip"),
       os.path.join(HERE, "csrc", "tbnn_mid.hip"), os.path.join(HERE, "csrc", "tbnn_tall.hip"),
       os.path.join(HERE, "csrc", "adapter.cpp")]
# both kernel families: the chain MFMAs write ArchVGPRs (their results feed the VALU), dW accumulators are pinned to
# AccVGPRs by hand (kernels_fast.hpp, mfma16_acc)
NARROW_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
TALL_NOP = ["-DTBNN_ASM_MFMA_NOP=0"]
PER_SOURCE_FLAGS = {"tbnn_api.hip": NARROW_FLAGS,
                    # experiments: TBNN_WIDE_FLAGS adds flags to the wide translation unit, TBNN_WIDE_AGPR_FORM=1 drops the VGPR form there
                    "tbnn_wide.hip": ([] if os.environ.get("TBNN_WIDE_AGPR_FORM") == "1" else NARROW_FLAGS)
                                     + os.environ.get("TBNN_WIDE_FLAGS", "").split(),
                    "tbnn_mid.hip": NARROW_FLAGS + os.environ.get("TBNN_MID_FLAGS", "").split(),
                    # the tall family's dW_0 runs thirteen 2-pass 4x4x1 MFMAs per k-step: nothing hides the two wait states an asm MFMA carries by
                    # default there (measured: -1.5 % at the tutorial shape), so this unit is built without them and the check in the compile
                    # repairs the pairs it finds (none in the shapes of the registry)
                    "tbnn_tall.hip": NARROW_FLAGS + TALL_NOP + os.environ.get("TBNN_TALL_FLAGS", "").split()}
# TBNN_BUILD_TAG=<tag>: a diagnostic variant (TBNN_EXTRA_FLAGS / TBNN_*_FLAGS) built side by side as libtbnn_<tag>.so with its
# own object directory; load it with TBNN_LIB=<path>
_TAG = os.environ.get("TBNN_BUILD_TAG", "")
OBJ_DIR = os.path.join(HERE, "_obj" + ("_" + _TAG if _TAG else ""))
OUT = os.path.join(HERE, "libtbnn" + ("_" + _TAG if _TAG else "") + ".so")


def _deps():
    d = list(SRC)
    cs = os.path.join(HERE, "csrc")
    d += [os.path.join(cs, f) for f in os.listdir(cs) if f.endswith(".hpp")]
    d.append(os.path.join(HERE, "..", "include", "tbnn.h"))
    d += [os.path.join(HERE, "hazard_lint.py"), os.path.join(HERE, "checked_compile.py")]      # the check is part of the compile
    return d


def sources_id(flags=()) -> str:
    """sha256 over everything the library is compiled from (and how): csrc/*, include/tbnn.h, the compiler flags"""
    import hashlib
    hsh = hashlib.sha256()
    cs = os.path.join(HERE, "csrc")
    for f in sorted(os.listdir(cs)):
        if f.endswith((".hpp", ".hip", ".cpp")):
            hsh.update(f.encode()); hsh.update(open(os.path.join(cs, f), "rb").read())
    hsh.update(open(os.path.join(HERE, "..", "include", "tbnn.h"), "rb").read())
    for f in ("hazard_lint.py", "checked_compile.py"):
        hsh.update(open(os.path.join(HERE, f), "rb").read())
    hsh.update(repr((list(flags), sorted(PER_SOURCE_FLAGS.items()))).encode())
    return hsh.hexdigest()[:16]


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in _deps()):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-value", "-Wno-unused-result"]
    flags += os.environ.get("TBNN_EXTRA_FLAGS", "").split()      # diagnostic builds (-DWIDE_DBG_...)
    bid = sources_id(flags)
    os.makedirs(OBJ_DIR, exist_ok=True)
    # one compile per translation unit, side by side (the kernel families take ~1 min each).  Every .hip unit goes through
    # checked_compile.run: hipcc's own steps replayed with the MFMA hazard check between the device listing and the assembler (wait states
    # inserted where a pair lacks them), the finished object disassembled and checked again.  No unit is linked unchecked.
    from concurrent.futures import ThreadPoolExecutor
    from . import checked_compile
    jobs, objs = [], []
    for src in SRC:
        obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        cmd = [hipcc] + flags + PER_SOURCE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if os.path.basename(src) == "tbnn_api.hip":
            cmd.insert(1, f'-DTBNN_BUILD_ID="{bid}"')
        if src.endswith(".hip"):
            cmd[1:1] = ["-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics"]   # per-kernel register / scratch remarks on stderr
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append(cmd)
        objs.append(obj)

    def one(cmd):
        if cmd[-3].endswith(".hip"):
            r = checked_compile.run(cmd, keep_listing=cmd[-1][:-2] + ".s" if os.environ.get("TBNN_KEEP_LISTING") == "1" else None)
            return r.rc, r.stderr, r.status
        p = subprocess.run(cmd, capture_output=True, text=True)
        return p.returncode, p.stderr, None

    with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
        results = list(pool.map(one, jobs))
    statuses = []
    for cmd, obj, (rc, err, status) in zip(jobs, objs, results):
        with open(obj + ".stderr", "w") as f:
            f.write(err)
        remarks = [l for l in err.splitlines() if "-Rpass-analysis=kernel-resource-usage" in l]
        other = [l for l in err.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l and not l.startswith("In file included from")
                 and "warnings generated" not in l and "warning generated" not in l and l.strip()]
        if other and (verbose or rc != 0):
            print("\n".join(other), file=sys.stderr, flush=True)
        if rc != 0:
            raise subprocess.CalledProcessError(rc, cmd)
        if status is not None:
            unit = os.path.basename(cmd[-3])
            statuses.append(f"{unit}: {status}")
            if verbose:
                print(f"{unit}: {status}", flush=True)
        # a FUSED kernel (one wave per SIMD, hand-planned register files) that needs scratch has lost its register plan:
        # accumulators demoted to a stack array are read back without the wait states an MFMA result needs
        name = None
        for l in remarks:
            if "Function Name:" in l:
                name = l.split("Function Name:")[1].split("[")[0].strip()
            elif "ScratchSize [bytes/lane]:" in l and name and any(k in name for k in ("k_fwd_bwd_", "k_chain_wide", "k_dw_wide", "k_forward_fast3")):
                if int(l.split("ScratchSize [bytes/lane]:")[1].split("[")[0]) > 0 and os.environ.get("TBNN_ALLOW_SPILL") != "1":   # (stamped diagnostic builds)
                    raise RuntimeError(f"{os.path.basename(cmd[-3])}: fused kernel {name} spills to scratch")
    # what the check did, linked into the library: tbnn_lint_status()
    lsrc = os.path.join(OBJ_DIR, "lint_status.cpp")
    with open(lsrc, "w") as f:
        off = [u for u, f in PER_SOURCE_FLAGS.items() if "-DTBNN_ASM_MFMA_NOP=0" in flags + f]
        statuses.append("wait states inside the asm MFMAs: on" + (f" (off in {', '.join(off)}: left to the check)" if off else ""))
        text = "; ".join(statuses).replace("\\", "/").replace('"', "'")
        f.write('extern "C" const char* tbnn_lint_status(void) { return "' + text + '"; }\n')
    lobj = os.path.join(OBJ_DIR, "lint_status.o")
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O1", "-fPIC", "-c", lsrc, "-o", lobj])
    objs.append(lobj)
    # link next to the target and rename: a rank that waits for the file (bench.py) never maps a half-written library
    tmp = OUT + f".{os.getpid()}.tmp"
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
