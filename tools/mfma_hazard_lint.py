"""dev / build check: inline-asm MFMAs whose A / B operand VGPR was written by a VALU instruction (v_accvgpr_read of a parked operand, a move, an
FMA ...) fewer than NEED wait states earlier.  The compiler inserts no software wait states around inline asm; the hardware needs them between a
VALU write of a VGPR and an MFMA that reads it as SrcA / SrcB (an MFMA issued right behind the write reads the OLD register: wrong, unrepeatable
results -- k_dw_wide at 15 -> 170 -> 114 -> 1, round 5).  Reads a `hipcc --cuda-device-only -S` listing:
  python tools/mfma_hazard_lint.py build/asm/api.s [NEED=2]
Wait states counted as the compiler's hazard recognizer does: every instruction between the write and the MFMA is one, `s_nop N` is N + 1."""
import re, sys
path = sys.argv[1]; NEED = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lines = open(path).read().split("\n")
kernel = None; in_asm = False
window = []            # recent instructions: (text, written vgprs or empty, wait states it provides)
found = {}
def vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
for ln in lines:
    t = ln.strip()
    m = re.match(r"^(_Z\w+):", ln)
    if m: kernel = m.group(1); window = []; continue
    if t.startswith(";;#ASMSTART"): in_asm = True; continue
    if t.startswith(";;#ASMEND"): in_asm = False; continue
    if not t or t.startswith((";", ".")) or t.endswith(":"): 
        if t.endswith(":") and not t.startswith(";"): window = []          # a label: another path may join here; be silent about what precedes it
        continue
    op = t.split()[0]
    args = [a.strip() for a in t[len(op):].split(";")[0].split(",")]
    if op.startswith("v_mfma") and in_asm:
        src = set()
        for a in args[1:3]: src |= vregs(a)
        dist = 0
        for (txt, wr, ws) in reversed(window):
            if wr & src and dist < NEED:
                found.setdefault(kernel, []).append((txt, t, dist))
                break
            dist += ws
            if dist >= NEED: break
    wr = set()
    if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_accvgpr_write")) and args:
        wr = vregs(args[0])
    ws = (int(args[0]) + 1) if op == "s_nop" and args and args[0].isdigit() else 1
    window.append((t, wr, ws)); window = window[-8:]
tot = sum(len(v) for v in found.values())
for k, v in found.items():
    print(f"{k[:110]}: {len(v)} asm MFMAs read a VGPR a VALU instruction wrote < {NEED} wait states earlier; first: `{v[0][0]}` -> `{v[0][1]}` ({v[0][2]} between)")
print(f"{path}: {tot} hazards in {len(found)} kernels")
sys.exit(1 if tot else 0)
