"""Build-time check of the compiled kernels: an MFMA must not read, as SrcA / SrcB, a VGPR that a VALU instruction wrote fewer than two wait
states earlier (gfx90a / gfx940 / gfx950: the hardware does not interlock this pair; an MFMA issued right behind the write reads the OLD
register).  The compiler keeps the distance for its own MFMAs; it inserts nothing around INLINE ASM, and the fused kernels' dW accumulation is
inline asm (accumulators pinned to AccVGPRs, kernels_fast.hpp: mfma16_acc).  When register pressure makes the compiler park an operand in an
AccVGPR, it comes back through `v_accvgpr_read` -- a VALU write -- possibly right in front of the asm MFMA: wrong and unrepeatable dW tiles
(k_dw_wide at 15 -> 170 -> 114 -> 1 with two waves per SIMD, found by tools/experiments/transition_fuzz.py in round 5).

build.py and jit.py disassemble every kernel object they produce and call `check()` (= `hazards_cfg`: the pairs along the control flow of the
disassembly, branch targets from the encoded offsets -- a pair may straddle a branch or a join); a unit with hazards is rebuilt with
-DTBNN_ASM_MFMA_NOP=1 (every asm MFMA carries its own two wait states) and checked again; one that still shows a pair (compiler-generated code:
an MFMA result moved at a join) is refused and the next kernel family takes the shape.

Wait states are counted as the compiler's hazard recognizer counts them: one per instruction between the write and the MFMA, N + 1 for `s_nop N`."""
import os
import re
import shutil
import subprocess
import tempfile

NEED = 2
LINEAR = bool(int(os.environ.get("TBNN_LINT_LINEAR", "0")))
SRCC = bool(int(os.environ.get("TBNN_LINT_SRCC", "0")))
WAW = bool(int(os.environ.get("TBNN_LINT_WAW", "0")))     # diagnostic: also report a VALU write of a register an MFMA in flight will write
LLVM_BIN = os.environ.get("TBNN_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def _vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


# The second pair: an MFMA writes ArchVGPRs (the chain MFMAs: -amdgpu-mfma-vgpr-form) and a VALU instruction reads them before the result has
# landed -- passes + 2 wait states (what the compiler keeps: 2-pass 4x4x1 4, 8-pass 16x16x4 f32 10, 16-pass 18).  The compiler keeps that distance for
# VALU code it generates; an INLINE-ASM reader (the packed relu mask / multiply of kernels_fast.hpp: relu_step2, pkmul2, mul_legacy) relies
# on mfma_settle() standing in between, and a reader that slips in front of it reads the old register (k_fwd_bwd_mid at 80 -> 80 -> 51 -> 2,
# round 5: delta of the last tile wrong by percent).
def _mfma_read_need(op: str) -> int:
    if "4x4x" in op:
        return 4
    if "32x32x" in op:
        return 18
    return 10


def hazards(listing: str, need: int = NEED, asm_only: bool = False):
    """[(kernel, writer, reader, wait states between)] in a disassembly (llvm-objdump -d) or a compiler listing (-S; asm_only: only the MFMAs
    between ;;#ASMSTART / ;;#ASMEND are checked for the first pair)"""
    out, kernel, window, in_asm = [], None, [], False
    mwrites = []          # MFMA results in ArchVGPRs not yet landed: [regs, text, wait states so far, needed]
    for ln in listing.split("\n"):
        t = ln.strip()
        m = re.match(r"^[0-9a-f]* ?<?(_Z\w+)>?:", ln)
        if m:
            kernel, window = m.group(1), []
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        if t.endswith(":"):
            window, mwrites = [], []          # a label: another path joins here
            continue
        t = t.split("//")[0].split(";")[0].strip()
        if not t:
            continue
        op = t.split()[0]
        args = [a.strip() for a in t[len(op):].split(",")]
        if op.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")):
            if not LINEAR or not op.startswith("s_cbranch"):
                window, mwrites = [], []      # a disassembly has no labels: what follows a branch may be reached from elsewhere
                continue                      # (LINEAR, diagnostic: the fall-through path of a conditional branch is followed: false positives possible)
        if op.startswith("v_mfma") and (in_asm or not asm_only):
            src = set()
            for a in (args[1:4] if SRCC else args[1:3]):      # (SRCC, diagnostic: the accumulator operand as well)
                src |= _vregs(a)
            dist = 0
            for txt, wr, ws in reversed(window):
                if wr & src:
                    out.append((kernel, txt, t, dist))
                    break
                dist += ws
                if dist >= need:
                    break
        wr = set()
        if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_accvgpr_write", "v_smfmac")) and args:
            wr = _vregs(args[0])
        ws = (int(args[0]) + 1) if op == "s_nop" and args and args[0].isdigit() else 1
        # second pair: a VALU (non-MFMA) instruction reading an MFMA result that has not landed
        if op.startswith("v_") and not op.startswith(("v_mfma", "v_smfmac")) and mwrites:
            srcs = set()
            for a in (args if op.startswith(("v_cmp", "v_accvgpr_write")) else args[1:]):
                srcs |= _vregs(a.split()[0] if a else a)
            for regs, txt, age, needed in mwrites:
                if regs & srcs and age < needed:
                    out.append((kernel, txt, t, age))
                    break
        if WAW and wr:
            for regs, txt, age, needed in mwrites:
                if regs & wr and age < needed:
                    out.append((kernel, txt, "WAW " + t, age))
                    break
        mwrites = [[r - wr, x, a + ws, nd] for r, x, a, nd in mwrites if a + ws < nd and (r - wr)]      # (a register written since is that writer's)
        if op.startswith("v_mfma") and args:
            dst = _vregs(args[0])
            if dst:
                mwrites.append([dst, t, 0, _mfma_read_need(op)])
        window.append((t, wr, ws))
        window = window[-(need + 2):]
    return out


def _parse_functions(listing: str):
    """llvm-objdump -d text -> {function: [(addr, op, args, text)]}"""
    funcs, cur = {}, None
    for ln in listing.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(_Z\w+)>:", ln)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None or "//" not in ln:
            continue
        code, _, cm = ln.partition("//")
        ma = re.match(r"\s*([0-9A-Fa-f]+):", cm)
        t = code.strip()
        if not ma or not t:
            continue
        op = t.split()[0]
        cur.append((int(ma.group(1), 16), op, [a.strip() for a in t[len(op):].split(",")], t))
    return funcs


def hazards_cfg(listing: str, need: int = NEED):
    """The two checks of `hazards` along the CONTROL FLOW of a disassembly (branch targets from the encoded offsets): a pair may straddle a
    branch or a join -- the compiler's own hazard recognizer has been seen to miss an MFMA at the end of a wave-uniform `if` block whose result
    a move at the join reads two instructions later (cooperative tail of k_fwd_bwd_fast3 at 13 -> 36 -> 16 -> 33 -> 32 -> 2, round 5)."""
    out = []
    for fn, ins in _parse_functions(listing).items():
        idx = {a: i for i, (a, _o, _g, _t) in enumerate(ins)}

        def succ(i):
            a, op, args, _t = ins[i]
            if op.startswith("s_endpgm"):
                return []
            if op.startswith(("s_branch", "s_cbranch")):
                off = int(args[0])
                if off >= 32768:
                    off -= 65536
                tgt = idx.get(a + 4 + 4 * off)
                nxt = [tgt] if tgt is not None else []
                if op.startswith("s_cbranch") and i + 1 < len(ins):
                    nxt.append(i + 1)
                return nxt
            return [i + 1] if i + 1 < len(ins) else []

        def ws_of(op, args):
            return (int(args[0]) + 1) if op == "s_nop" and args and args[0].isdigit() else 1

        def valu(op):
            return op.startswith("v_") and not op.startswith(("v_mfma", "v_smfmac"))

        def reads(op, args):
            r = set()
            for a in (args if op.startswith(("v_cmp", "v_accvgpr_write")) else args[1:]):
                r |= _vregs(a.split()[0] if a else a)
            return r

        def writes(op, args):
            if op.startswith("v_") and not op.startswith(("v_cmp", "v_accvgpr_write")) and args:
                return _vregs(args[0])
            return set()

        for i, (a, op, args, t) in enumerate(ins):
            # (1) VALU write -> MFMA SrcA / SrcB read
            if valu(op):
                wr = writes(op, args)
                if wr:
                    stack, seen = [(j, 0) for j in succ(i)], set()
                    while stack:
                        j, age = stack.pop()
                        if age >= need or (j, age) in seen:
                            continue
                        seen.add((j, age))
                        _a, o2, g2, t2 = ins[j]
                        if o2.startswith("v_mfma"):
                            src = set()
                            for x in g2[1:3]:
                                src |= _vregs(x)
                            if src & wr:
                                out.append((fn, t, t2, age))
                                break
                        if writes(o2, g2) >= wr and not o2.startswith("v_mfma"):
                            continue
                        stack += [(k, age + ws_of(o2, g2)) for k in succ(j)]
            # (2) MFMA result (ArchVGPRs) -> VALU read before it has landed
            if op.startswith("v_mfma") and args:
                dst = _vregs(args[0])
                if not dst:
                    continue
                needed = _mfma_read_need(op)
                stack, seen, hit = [(j, 0, frozenset(dst)) for j in succ(i)], set(), False
                while stack and not hit:
                    j, age, regs = stack.pop()
                    if age >= needed or not regs or (j, age, regs) in seen:
                        continue
                    seen.add((j, age, regs))
                    _a, o2, g2, t2 = ins[j]
                    if valu(o2) and reads(o2, g2) & regs:
                        out.append((fn, t, t2, age))
                        hit = True
                        break
                    regs2 = regs - writes(o2, g2) if not o2.startswith("v_mfma") else regs - _vregs(g2[0]) if g2 else regs
                    stack += [(k, age + ws_of(o2, g2), frozenset(regs2)) for k in succ(j)]
    return out


def disassemble(path: str) -> str:
    """device code (gfx950) of a fat object / shared library as llvm-objdump text"""
    d = tempfile.mkdtemp(prefix="tbnn_lint_")
    try:
        f = os.path.join(d, os.path.basename(path))
        shutil.copy(path, f)
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", f], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        texts = []
        for g in sorted(os.listdir(d)):
            if "amdgcn" in g:
                texts.append(subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", os.path.join(d, g)], check=True, capture_output=True,
                                            text=True).stdout)
        if not texts:
            raise RuntimeError(f"no device code object found in {path}")
        return "\n".join(texts)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def check(path: str, need: int = NEED):
    """hazards of a compiled object / library (empty list: clean), along its control flow"""
    return hazards_cfg(disassemble(path), need)


def describe(found, limit: int = 3) -> str:
    per = {}
    for k, w, m, dist in found:
        per.setdefault(k, []).append((w, m, dist))
    return "; ".join(f"{k[:90]}: {len(v)} (first: `{v[0][0]}` -> `{v[0][1]}`, {v[0][2]} wait states)" for k, v in list(per.items())[:limit])


if __name__ == "__main__":
    import sys
    p = sys.argv[1]
    need = int(sys.argv[2]) if len(sys.argv) > 2 else NEED
    found = hazards(open(p).read(), need, asm_only=True) if p.endswith(".s") else check(p, need)
    print(describe(found, 20) if found else "clean")
    print(f"{p}: {len(found)} hazards")
    sys.exit(1 if found else 0)
