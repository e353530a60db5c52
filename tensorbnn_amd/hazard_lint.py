"""Build-time check of the compiled kernels for the data hazards gfx950 leaves to software around its matrix instructions.

The hardware does not interlock an MFMA against the vector instructions around it; the pairs below need WAIT STATES (one per instruction issued
between the two, N + 1 for `s_nop N`) and read an OLD register without them.  The compiler's hazard recognizer keeps the distances for code it
generates -- but it sees nothing INSIDE an inline-asm statement, and the fused kernels' dW accumulation, packed relu masks and LDS operand loads
are inline asm (kernels_fast.hpp: mfma16_acc, mfma4_acc, relu_step2, pkmul2; kernels_fast3.hpp: nf_load / nf_mfma); it has also been seen to
miss a pair across a join of the control flow (round 5).  Every pair with one side inside an asm string is therefore checked HERE, on the
disassembly of every unit build.py / jit.py produce, along the control flow (branch targets from the encoded offsets).

Rules and their numbers.  They are the ones LLVM's GCNHazardRecognizer applies to gfx940 / gfx950 (llvm/lib/Target/AMDGPU/GCNHazardRecognizer.cpp:
checkMAIHazards90A, checkMAIVALUHazards and their GFX940_SMFMA_N_Pass* helpers; the f32 MFMAs of this repo are "SGEMM", not "XDL", instructions),
and tests/test_host_logic.py::test_hazard_rules_agree_with_the_installed_compiler re-derives every one of them from the `s_nop`s the installed
llc puts into probe kernels.  P = passes of the producing MFMA (v_mfma_f32_4x4x1: 2, 16x16x1 / 16x16x4: 8, 32x32x1 / 32x32x2: 16; any other
matrix instruction is priced at the worst case of the table, 16 passes and the XDL increments):

  R1  valu->mfma     a VALU instruction (v_accvgpr_read / _write / _mov included) writes a VGPR or AccVGPR, an MFMA reads it as SrcA, SrcB or
                     SrcC: 2 wait states                                                  (checkMAIHazards90A: LegacyVALUWritesVGPRWaitStates)
  R2a mfma->srcab    an MFMA's result is read as SrcA / SrcB of a later MFMA: P + 2          (GFX940_SMFMA_N_PassWritesVGPROverlappedSrcABWaitStates)
  R2b mfma->srcc     ... as SrcC: the SAME registers (back-to-back accumulation): 2 after a 2-pass MFMA, else 0; OVERLAPPING but different
                     registers: P                                                           (GFX940_SMFMA4x4WritesVGPRFullSrcCWaitStates, ..OverlappedSMFMASrcC..)
  R2c mfma->valu     ... read OR overwritten by a VALU instruction: P + 2                   (GFX940_SMFMA_N_PassWriteVgprVALUMemExpReadWaitStates, ..VALUWaw..)
  R2d mfma->mem      ... used as data or address of an LDS / global / buffer / flat / scratch / export instruction: P + 2        (same helper)
  R3  exec->mfma     a VALU instruction writes EXEC (v_cmpx), an MFMA follows: 4              (checkMAIHazards90A: VALUWritesExecWaitStates)
  R5  trans->valu    a transcendental instruction (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos) writes a VGPR, a VALU instruction that is
                     not itself transcendental reads it: 1 wait state     (checkVALUHazards: TransDefWaitstates, hasTransForwardingHazard = gfx940+).
                     The kernels keep their inline-asm packed instructions away from such results at the source (MidCfg's PKA); this is the net under that.
  R4  load->use      a register loaded by an LDS or vector-memory instruction is touched before an `s_waitcnt` that covers the load.  The
                     compiler counts its own loads; a load issued from an asm string (nf_load's `ds_read_b128 a[..]`) is invisible to it, and
                     a copy or spill the register allocator puts between that load and the hand-written wait would read the register early.
                     Counters as SIInsertWaitcnts models gfx9: LDS operations return in order among themselves, vector-memory operations
                     likewise (loads and stores share vmcnt); scalar loads may overtake, so only LDS operations issued after the load count
                     towards `lgkmcnt(n)`; a FLAT access needs both counters at zero.

`check()` returns a list of findings (kernel, producer, consumer, wait states seen, rule, wait states needed, the two addresses) -- empty: clean.  A unit with
findings is rebuilt with -DTBNN_ASM_MFMA_NOP=1 (every asm MFMA carries its own wait states) and checked again; what still shows a pair is refused.
When no disassembler is found the unit cannot be checked: build.py / jit.py then compile it with the wait states in and say so (RuntimeWarning,
`lint_status()`), they never hand out an unchecked unit."""
import os
import re
import shutil
import subprocess
import tempfile

NEED = 2                     # R1
EXEC_NEED = 4                # R3
MAX_PASSES = 16
LINEAR = bool(int(os.environ.get("TBNN_LINT_LINEAR", "0")))
RULES_OFF = set(filter(None, os.environ.get("TBNN_LINT_OFF", "").split(",")))      # diagnostic: e.g. TBNN_LINT_OFF=R4


def llvm_bin() -> str:
    """directory of llvm-objdump: TBNN_LLVM_BIN, else next to the hipcc in use (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), else /opt/rocm's"""
    env = os.environ.get("TBNN_LLVM_BIN")
    if env:
        return env
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cands = []
    real = os.path.realpath(shutil.which(hipcc) or hipcc)
    for h in (hipcc, real):
        root = os.path.dirname(os.path.dirname(os.path.abspath(h)))
        cands += [os.path.join(root, "lib", "llvm", "bin"), os.path.join(root, "llvm", "bin")]
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")):
            return c
    return cands[0]


LLVM_BIN = llvm_bin()


def available() -> bool:
    return os.path.exists(os.path.join(LLVM_BIN, "llvm-objdump"))


# ---- registers: v<n> -> n, a<n> -> 1024 + n (gfx90a and later: one unified file; an MFMA takes either kind for every operand)
_REG = re.compile(r"(?<![\w\]])([va])(?:(\d+)\b|\[(\d+):(\d+)\])")
AOFF = 1024


def _regs(tok: str):
    """VGPRs / AccVGPRs named in one operand, modifiers (-v1, |v1|, neg(v1), sext(v1), `v1 dst_sel:..`) stripped"""
    out = set()
    for kind, one, lo, hi in _REG.findall(tok):
        base = AOFF if kind == "a" else 0
        if one:
            out.add(base + int(one))
        else:
            out.update(range(base + int(lo), base + int(hi) + 1))
    return out


def _vregs(tok):        # (kept for callers of the first version: ArchVGPRs of one operand)
    return {r for r in _regs(tok) if r < AOFF}


def _fmt(r):
    return f"a{r - AOFF}" if r >= AOFF else f"v{r}"


_MEM = ("ds_", "global_", "buffer_", "tbuffer_", "flat_", "scratch_", "image_", "exp")
_VM = ("global_", "buffer_", "tbuffer_", "scratch_", "image_")
_TWO_DST = ("v_swap_b32", "v_permlane16_swap", "v_permlane32_swap")


def is_mfma(op):
    return op.startswith(("v_mfma", "v_smfmac"))


def is_valu(op):
    return op.startswith("v_") and not is_mfma(op)


def is_mem(op):
    return op.startswith(_MEM)


_TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)(_legacy)?_(f16|f32|bf16)")


def is_trans(op):
    return _TRANS.match(op) is not None


def mfma_passes(op: str) -> int:
    """passes of an f32 (SGEMM) MFMA; anything else: the table's worst case"""
    m = re.match(r"v_mfma_f32_(\d+)x(\d+)x(\d+)(?:_\d+b)?_f32", op)
    if m:
        return {4: 2, 16: 8, 32: 16}.get(int(m.group(1)), MAX_PASSES)
    return MAX_PASSES


def is_sgemm(op: str) -> bool:
    return re.match(r"v_mfma_f32_\d+x\d+x\d+(?:_\d+b)?_f32", op) is not None


def need_read(op: str) -> int:
    """R2a / R2c / R2d: wait states before the result of MFMA `op` may be touched by a VALU / memory instruction or read as SrcA / SrcB"""
    p = mfma_passes(op)
    return p + 2 if is_sgemm(op) else p + 3 + (1 if p != 2 else 0)


def need_srcc(op: str, same: bool) -> int:
    """R2b: ... read as SrcC by the next MFMA; `same`: exactly the producer's destination registers"""
    p = mfma_passes(op)
    if same:
        return 2 if p == 2 else 0
    return p if is_sgemm(op) else p + 2


_mfma_read_need = need_read


class Ins:
    __slots__ = ("addr", "op", "args", "text", "enc", "rd", "wr", "mf", "src_ab", "src_c", "dst", "ws", "wexec", "line")

    def __init__(self, addr, text, enc=None, line=-1):
        self.addr, self.text, self.enc, self.line = addr, text, enc, line
        op = self.op = text.split()[0]
        args = self.args = [a.strip() for a in text[len(op):].split(",")] if len(text) > len(op) else []
        self.ws = (int(args[0]) + 1) if op == "s_nop" and args and args[0].isdigit() else 3 if op.startswith("s_swappc") else 1      # (a call: the callee's entry wait and return at least)
        self.mf = is_mfma(op)
        self.src_ab, self.src_c, self.dst, self.wexec = set(), set(), set(), False
        rd, wr = set(), set()
        if self.mf:
            self.dst = wr = _regs(args[0]) if args else set()
            for a in args[1:3]:
                self.src_ab |= _regs(a)
            if op.startswith("v_smfmac"):
                self.src_c = set(self.dst)                      # the sparse forms accumulate into vdst; operand 3 is the index register
                if len(args) > 3:
                    self.src_ab |= _regs(args[3])
            elif len(args) > 3:
                self.src_c = _regs(args[3])
            rd = self.src_ab | self.src_c
        elif is_valu(op):
            if op.startswith("v_cmpx") or (args and args[0] == "exec"):
                self.wexec = True
            ndst = 2 if op.startswith(_TWO_DST) else 1
            if op.startswith("v_cmp"):
                ndst = 0                                        # SGPR / VCC / EXEC destinations
            for a in args[:ndst]:
                wr |= _regs(a)
            for a in args[ndst:]:
                rd |= _regs(a)
            if ndst == 2:
                rd |= wr
        elif is_mem(op):
            store = any(k in op for k in ("store", "ds_write", "ds_add", "ds_sub", "ds_min", "ds_max", "ds_and", "ds_or", "ds_xor", "ds_inc", "ds_dec",
                                           "ds_cmpst", "ds_wrxchg", "ds_mskor", "ds_gws", "ds_nop", "ds_append", "ds_consume", "atomic")) or op == "exp"
            returns = "_rtn" in op or (("atomic" in op) and (" glc" in text or " sc0" in text))
            lds_dma = "_lds_" in op or re.search(r"\blds\b", text[len(op):]) is not None      # LDS-DMA loads: every VGPR operand is an address
            if (store and not returns) or lds_dma:
                for a in args:
                    rd |= _regs(a)
            else:
                if args:
                    wr |= _regs(args[0])
                for a in args[1:]:
                    rd |= _regs(a)
                if "d16" in op or " tfe" in text or returns:    # partial / extra writes: the destination counts as read as well
                    rd |= wr
        self.rd, self.wr = rd, wr


def _parse_functions(listing: str):
    """llvm-objdump -d text, or a compiler listing (-S: no addresses, instructions numbered by position, branch operands are `.LBBn_m`
    labels) -> {function: [Ins]}"""
    funcs, cur, labels, pend = {}, None, {}, []
    n = 0
    for lineno, ln in enumerate(listing.split("\n")):
        if ln and not ln[0].isspace():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln) or re.match(r"^([A-Za-z_$][\w.$]*):", ln)
            if m:
                cur = funcs.setdefault(m.group(1), [])
                pend = []
                continue
            ml = re.match(r"^(\.L[\w.$]+):", ln)
            if ml:
                pend.append(ml.group(1))
                continue
        t = ln.strip()
        if cur is None or not t or t.startswith((";", ".", "//")):
            continue
        code, _, cm = ln.partition("//")
        code = code.split(";")[0].strip()
        if not code or code.endswith(":"):
            continue
        ma = re.match(r"\s*([0-9A-Fa-f]+):\s*([0-9A-Fa-f]{8})?", cm)
        if ma:
            addr, enc = int(ma.group(1), 16), (int(ma.group(2), 16) if ma.group(2) else None)
        else:
            addr, enc = 4 * n, None
        n += 1
        ins = Ins(addr, code, enc, lineno)
        for lb in pend:
            labels[lb] = ins
        pend = []
        cur.append(ins)
    for f in funcs.values():
        for k, ins in enumerate(f):
            if ins.op.startswith(("s_branch", "s_cbranch")) and ins.args and ins.args[0] in labels:
                ins.enc = ("label", labels[ins.args[0]].addr)
            elif ins.op.startswith("s_setpc") and k >= 2:
                # a relaxed long branch: s_getpc_b64 s[n:n+1] / .Lpost_getpc: / s_add_u32 sn, sn, (.LBBx_y-.Lpost_getpc)&4294967295 / s_addc_u32 / s_setpc_b64
                for back in f[max(0, k - 4):k]:
                    mm = re.search(r"\(?(\.LBB\w+)-\(?\.Lpost_getpc", back.text)
                    if mm and mm.group(1) in labels:
                        ins.enc = ("label", labels[mm.group(1)].addr)
    return {k: v for k, v in funcs.items() if v}


def _successors(ins_list, unfollowed=None):
    idx = {x.addr: i for i, x in enumerate(ins_list)}
    succ = []
    for i, x in enumerate(ins_list):
        op = x.op
        nxt = [i + 1] if i + 1 < len(ins_list) else []
        if op.startswith(("s_endpgm", "s_trap")):
            nxt = []
        elif op.startswith("s_setpc"):
            nxt = []
            tgt = None
            if isinstance(x.enc, tuple):
                tgt = idx.get(x.enc[1])
            elif i >= 3:
                # disassembly of a relaxed long branch: s_getpc_b64 s[n:n+1]; s_add_u32 sn, sn, <lo>; s_addc_u32 sn+1, sn+1, <hi>; s_setpc_b64 s[n:n+1]
                g, a = ins_list[i - 3], ins_list[i - 2]
                if g.op.startswith("s_getpc") and a.op == "s_add_u32" and len(a.args) == 3 and re.match(r"^(0x[0-9a-fA-F]+|-?\d+)$", a.args[2]):
                    off = int(a.args[2], 0) & 0xFFFFFFFF
                    if off >= 1 << 31:
                        off -= 1 << 32
                    tgt = idx.get(a.addr + off)
            if tgt is not None:
                nxt = [tgt]
            elif unfollowed is not None and x.args and x.args[0] != "s[30:31]":
                unfollowed.append(x)                             # (s[30:31]: a function's return)
        elif op.startswith(("s_branch", "s_cbranch")):
            tgt = None
            if isinstance(x.enc, tuple):
                tgt = idx.get(x.enc[1])
            else:
                off = None
                if x.enc is not None:
                    off = x.enc & 0xFFFF                      # SOPP: simm16 in the low half of the encoded word (whatever the operand text looks like)
                elif x.args and re.match(r"^-?\d+$", x.args[0]):
                    off = int(x.args[0]) & 0xFFFF
                if off is not None:
                    if off >= 32768:
                        off -= 65536
                    tgt = idx.get(x.addr + 4 + 4 * off)
                if tgt is None and unfollowed is not None and not LINEAR:
                    unfollowed.append(x)
            if LINEAR:
                tgt = None
            fall = nxt if op.startswith("s_cbranch") else []
            nxt = ([tgt] if tgt is not None else []) + fall
        succ.append(nxt)
    return succ


def _waitcnt(ins):
    """(vmcnt, lgkmcnt) an s_waitcnt waits for; None: that counter is not waited on"""
    t = ins.text
    if ins.op.startswith("s_swappc"):
        return (0, 0)            # a call: every non-kernel function opens with s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0) (SIInsertWaitcnts)
    if ins.op == "s_waitcnt":
        vm = re.search(r"vmcnt\((\d+)\)", t)
        lg = re.search(r"lgkmcnt\((\d+)\)", t)
        if vm or lg or "expcnt" in t:
            return (int(vm.group(1)) if vm else None, int(lg.group(1)) if lg else None)
        m = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", t)
        if m:                                                     # raw immediate, gfx9 layout: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]
            v = int(m.group(1), 0)
            vmc, lgc = (v & 15) | ((v >> 14) & 3) << 4, (v >> 8) & 15
            return (None if vmc == 63 else vmc, None if lgc == 15 else lgc)
    return (None, None)


def _scan_function(fn, ins, out):
    lost = []
    succ = _successors(ins, lost)
    for x in lost:                                                # a branch whose target is not an instruction of this function: nothing behind it is checked
        out.append((fn, x.text, "(target not found)", 0, "R0 cfg", 0, x.addr, x.addr, x.line, x.line))
    n = len(ins)
    for i in range(n):
        x = ins[i]
        # (every path is followed to its first consumer; per consumer the SHORTEST distance is reported, so that one repair covers all paths)
        # ---- R1: VALU write -> MFMA operand read
        if "R1" not in RULES_OFF and is_valu(x.op) and x.wr:
            stack, seen, hits = [(j, 0, frozenset(x.wr)) for j in succ[i]], set(), {}
            while stack:
                j, age, regs = stack.pop()
                if age >= NEED or not regs or (j, age, regs) in seen:
                    continue
                seen.add((j, age, regs))
                y = ins[j]
                if y.mf and (y.rd & regs):
                    hits[j] = min(hits.get(j, age), age)
                    continue
                regs2 = regs - y.wr
                stack += [(k, age + y.ws, regs2) for k in succ[j]]
            for j, age in sorted(hits.items()):
                out.append((fn, x.text, ins[j].text, age, "R1 valu->mfma", NEED, x.addr, ins[j].addr, x.line, ins[j].line))
        # ---- R5: transcendental result -> non-transcendental VALU read
        if "R5" not in RULES_OFF and is_trans(x.op) and x.wr:
            for j in succ[i]:
                y = ins[j]
                if y.op.startswith("v_") and not is_trans(y.op) and (y.rd & x.wr):
                    out.append((fn, x.text, y.text, 0, "R5 trans->valu", 1, x.addr, y.addr, x.line, y.line))
        # ---- R3: VALU write of EXEC -> MFMA
        if "R3" not in RULES_OFF and x.wexec:
            stack, seen, hits = [(j, 0) for j in succ[i]], set(), {}
            while stack:
                j, age = stack.pop()
                if age >= EXEC_NEED or (j, age) in seen:
                    continue
                seen.add((j, age))
                y = ins[j]
                if y.mf:
                    hits[j] = min(hits.get(j, age), age)
                    continue
                stack += [(k, age + y.ws) for k in succ[j]]
            for j, age in sorted(hits.items()):
                out.append((fn, x.text, ins[j].text, age, "R3 exec->mfma", EXEC_NEED, x.addr, ins[j].addr, x.line, ins[j].line))
        # ---- R2: MFMA result -> anything that touches it before it has landed
        if "R2" not in RULES_OFF and x.mf and x.dst:
            nmax = need_read(x.op)
            stack, seen, hits = [(j, 0, frozenset(x.dst)) for j in succ[i]], set(), {}
            while stack:
                j, age, regs = stack.pop()
                if age >= nmax or not regs or (j, age, regs) in seen:
                    continue
                seen.add((j, age, regs))
                y = ins[j]
                hit = None
                if y.mf:
                    if y.src_ab & regs:
                        hit = ("R2a mfma->srcab", nmax)
                    elif y.src_c & regs:
                        nd = need_srcc(x.op, y.src_c == x.dst)
                        if age < nd:
                            hit = ("R2b mfma->srcc", nd)
                elif is_valu(y.op):
                    if (y.rd | y.wr) & regs:
                        hit = ("R2c mfma->valu", nmax)
                elif is_mem(y.op):
                    if (y.rd | y.wr) & regs:
                        hit = ("R2d mfma->mem", nmax)
                if hit:
                    if j not in hits or age < hits[j][0]:
                        hits[j] = (age,) + hit
                    continue
                regs2 = regs - y.wr
                stack += [(k, age + y.ws, regs2) for k in succ[j]]
            for j, (age, rule, nd) in sorted(hits.items()):
                out.append((fn, x.text, ins[j].text, age, rule, nd, x.addr, ins[j].addr, x.line, ins[j].line))
        # ---- R4: loaded register touched before a wait that covers the load
        if "R4" not in RULES_OFF and is_mem(x.op) and x.wr and not x.op.startswith("exp"):
            lds, flat = x.op.startswith("ds_"), x.op.startswith("flat_")
            # state: (instruction, in-order operations of the load's own counter issued since, registers still pending)
            stack, best = [(j, 0, frozenset(x.wr)) for j in succ[i]], {}
            steps = 0
            while stack:
                j, k, regs = stack.pop()
                if not regs:
                    continue
                key = (j, regs)
                if key in best and best[key] <= k:
                    continue
                best[key] = k
                steps += 1
                if steps > 200000:
                    break
                y = ins[j]
                vm, lg = _waitcnt(y)
                if flat:
                    done = vm == 0 and lg == 0
                elif lds:
                    done = lg is not None and lg <= k
                else:
                    done = vm is not None and vm <= k
                if done:
                    continue
                same_class = is_mem(y.op) and not flat and not y.op.startswith(("flat_", "exp")) and y.op.startswith("ds_") == lds
                if (y.rd & regs) or ((y.wr & regs) and not same_class):
                    out.append((fn, x.text, y.text, k, "R4 load->use", 0, x.addr, y.addr, x.line, y.line))
                    break
                regs = regs - y.wr                               # a later load of the same in-order counter owns them now
                if y.op.startswith(("s_endpgm", "s_setpc")):
                    continue
                k2 = k
                if lds and y.op.startswith("ds_"):
                    k2 = min(k + 1, 16)
                elif not lds and not flat and y.op.startswith(_VM):
                    k2 = min(k + 1, 64)
                stack += [(s, k2, regs) for s in succ[j]]


def hazards_cfg(listing: str, need: int = NEED):
    """every rule above, along the control flow of a disassembly: [(kernel, producer, consumer, wait states between, rule, wait states needed)]"""
    global NEED
    keep, NEED = NEED, need
    try:
        out = []
        for fn, ins in _parse_functions(listing).items():
            _scan_function(fn, ins, out)
        return out
    finally:
        NEED = keep


def hazards(listing: str, need: int = NEED, asm_only: bool = False):
    """the same rules on a straight-line reading of the listing (branches end a path: nothing is followed into a target); for `-S` listings
    `asm_only` keeps the findings whose MFMA stands between ;;#ASMSTART / ;;#ASMEND"""
    global LINEAR
    asm_lines = set()
    if asm_only:
        inside = False
        for ln in listing.split("\n"):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                inside = True
            elif t.startswith(";;#ASMEND"):
                inside = False
            elif inside and t.startswith(("v_mfma", "v_smfmac")):
                asm_lines.add(t.split(";")[0].split("//")[0].strip())
    keep, LINEAR = LINEAR, True
    try:
        found = hazards_cfg(listing, need)
    finally:
        LINEAR = keep
    if asm_only:
        found = [f for f in found if f[1] in asm_lines or f[2] in asm_lines]
    return found


def fix_listing(listing: str, need: int = NEED, rounds: int = 6):
    """A compiler listing (-S) with every finding repaired IN PLACE: the missing wait states (`s_nop`) -- for R4 a full `s_waitcnt` of the load's
    counter -- inserted in front of the consuming instruction, then checked again until nothing is left.  Returns (new listing, [findings that
    were repaired], [findings that remain]).  What remains (R0: a branch the scan could not follow) has no local repair."""
    fixed = []
    for _ in range(rounds):
        found = hazards_cfg(listing, need)
        todo = {}
        for f in found:
            rule = f[4].split()[0]
            if rule == "R0" or f[9] < 0:
                continue
            if rule == "R4":
                op = f[1].split()[0]
                what = "s_waitcnt vmcnt(0) lgkmcnt(0)" if op.startswith("flat_") else "s_waitcnt lgkmcnt(0)" if op.startswith("ds_") else "s_waitcnt vmcnt(0)"
                todo.setdefault(f[9], set()).add(what)
            else:
                miss = f[5] - f[3]
                cur = todo.setdefault(f[9], set())
                old = max([int(w.split("#")[1]) for w in cur if w.startswith("nop#")], default=0)
                cur.discard(f"nop#{old}")
                cur.add(f"nop#{max(old, miss)}")
            fixed.append(f)
        if not todo:
            return listing, fixed, found
        lines = listing.split("\n")
        for ln in sorted(todo, reverse=True):
            ins = []
            for w in sorted(todo[ln]):
                if w.startswith("nop#"):
                    k = int(w.split("#")[1])
                    while k > 0:
                        ins.append(f"\ts_nop {min(k, 16) - 1}")
                        k -= min(k, 16)
                else:
                    ins.append("\t" + w)
            lines[ln:ln] = ins
        listing = "\n".join(lines)
    return listing, fixed, hazards_cfg(listing, need)


def disassemble(path: str) -> str:
    """device code (gfx950) of a fat object / shared library as llvm-objdump text"""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    if not os.path.exists(objdump):
        raise FileNotFoundError(f"{objdump} not found (set TBNN_LLVM_BIN)")
    d = tempfile.mkdtemp(prefix="tbnn_lint_")
    try:
        f = os.path.join(d, os.path.basename(path))
        shutil.copy(path, f)
        subprocess.run([objdump, "--offloading", f], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        texts = []
        for g in sorted(os.listdir(d)):
            if "amdgcn" in g:
                texts.append(subprocess.run([objdump, "-d", os.path.join(d, g)], check=True, capture_output=True, text=True).stdout)
        if not texts:
            raise RuntimeError(f"no device code object found in {path}")
        return "\n".join(texts)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def check(path: str, need: int = NEED):
    """findings of a compiled object / library (empty list: clean), along its control flow"""
    return hazards_cfg(disassemble(path), need)


def describe(found, limit: int = 3) -> str:
    per = {}
    for f in found:
        per.setdefault(f[0], []).append(f)
    parts = []
    for k, v in list(per.items())[:limit]:
        rules = sorted({f[4].split()[0] for f in v if len(f) > 4})
        parts.append(f"{k[:90]}: {len(v)}{' ' + '/'.join(rules) if rules else ''} (first: `{v[0][1]}` -> `{v[0][2]}`, {v[0][3]} wait states"
                     + (f", {v[0][5]} needed" if len(v[0]) > 5 and v[0][5] else "") + ")")
    return "; ".join(parts)


if __name__ == "__main__":
    import sys
    p = sys.argv[1]
    need = int(sys.argv[2]) if len(sys.argv) > 2 else NEED
    if p.endswith((".s", ".dis", ".txt")):
        found = hazards_cfg(open(p).read(), need)
    else:
        found = check(p, need)
    print(describe(found, 20) if found else "clean")
    tally = {}
    for f in found:
        tally[f[4]] = tally.get(f[4], 0) + 1
    print(f"{p}: {len(found)} findings {tally if tally else ''}")
    sys.exit(1 if found else 0)
