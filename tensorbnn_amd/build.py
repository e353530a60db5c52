"""Build libtbnn.so in-tree with hipcc for gfx950 (`python -m tensorbnn_amd.build`)."""
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
PER_SOURCE_FLAGS = {"tbnn_api.hip": NARROW_FLAGS,
                    # experiments: TBNN_WIDE_FLAGS adds flags to the wide translation unit, TBNN_WIDE_AGPR_FORM=1 drops the VGPR form there
                    "tbnn_wide.hip": ([] if os.environ.get("TBNN_WIDE_AGPR_FORM") == "1" else NARROW_FLAGS)
                                     + os.environ.get("TBNN_WIDE_FLAGS", "").split(),
                    "tbnn_mid.hip": NARROW_FLAGS + os.environ.get("TBNN_MID_FLAGS", "").split(),
                    "tbnn_tall.hip": NARROW_FLAGS + os.environ.get("TBNN_TALL_FLAGS", "").split()}
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
    # one hipcc per translation unit, side by side (the two kernel families take ~1 min each)
    procs, objs = [], []
    for src in SRC:
        obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        cmd = [hipcc] + flags + PER_SOURCE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if os.path.basename(src) == "tbnn_api.hip":
            cmd.insert(1, f'-DTBNN_BUILD_ID="{bid}"')
        if src.endswith(".hip"):
            cmd[1:1] = ["-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics"]   # per-kernel register / scratch remarks on stderr
        if verbose:
            print(" ".join(cmd), flush=True)
        # stderr to a file per compile: the resource-usage remarks of dozens of kernels overflow a pipe, and a compiler blocked on
        # a full pipe until its turn to be drained would serialise the side-by-side build
        errf = open(obj + ".stderr", "w+")
        procs.append((cmd, subprocess.Popen(cmd, stderr=errf, text=True), errf))
        objs.append(obj)
    for cmd, p, errf in procs:
        p.wait()
        errf.seek(0); err = errf.read(); errf.close()
        remarks = [l for l in err.splitlines() if "-Rpass-analysis=kernel-resource-usage" in l]
        other = [l for l in err.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l and not l.startswith("In file included from")
                 and "warnings generated" not in l and "warning generated" not in l]
        if other and verbose:
            print("\n".join(other), file=sys.stderr, flush=True)
        if p.returncode != 0:
            print("\n".join(other), file=sys.stderr, flush=True)
            raise subprocess.CalledProcessError(p.returncode, cmd)
        # a FUSED kernel (one wave per SIMD, hand-planned register files) that needs scratch has lost its register plan:
        # accumulators demoted to a stack array are read back without the wait states an MFMA result needs
        name = None
        for l in remarks:
            if "Function Name:" in l:
                name = l.split("Function Name:")[1].split("[")[0].strip()
            elif "ScratchSize [bytes/lane]:" in l and name and any(k in name for k in ("k_fwd_bwd_", "k_chain_wide", "k_dw_wide", "k_forward_fast3")):
                if int(l.split("ScratchSize [bytes/lane]:")[1].split("[")[0]) > 0 and os.environ.get("TBNN_ALLOW_SPILL") != "1":   # (stamped diagnostic builds)
                    raise RuntimeError(f"{os.path.basename(cmd[-3])}: fused kernel {name} spills to scratch")
    # every kernel object disassembled and checked for VALU-write -> asm-MFMA-read pairs without wait states (hazard_lint.py); a unit that shows
    # one is rebuilt with the wait states inside the asm statements and must then be clean
    from . import hazard_lint
    for (cmd, _p, _e), obj in zip(procs, objs):
        if not cmd[-3].endswith(".hip"):
            continue
        try:
            found = hazard_lint.check(obj)
        except (OSError, subprocess.CalledProcessError, RuntimeError) as e:      # no disassembler on this machine: built as it is, said so
            print(f"{os.path.basename(obj)}: not checked for MFMA operand hazards ({e})", file=sys.stderr, flush=True)
            continue
        if found:
            print(f"{os.path.basename(obj)}: {len(found)} asm MFMAs behind a VALU write of their operand ({hazard_lint.describe(found, 2)}): "
                  "rebuilding with -DTBNN_ASM_MFMA_NOP=1", file=sys.stderr, flush=True)
            cmd2 = cmd[:1] + ["-DTBNN_ASM_MFMA_NOP=1"] + cmd[1:]
            subprocess.run(cmd2, check=True, stderr=subprocess.DEVNULL)
            found = hazard_lint.check(obj)
            if found:
                raise RuntimeError(f"{os.path.basename(obj)}: MFMA operand hazards remain: {hazard_lint.describe(found)}")
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
