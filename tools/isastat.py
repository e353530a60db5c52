"""dev: instruction mix of one kernel's MAIN LOOP in a gfx950 assembly listing (hipcc --cuda-device-only -S):
  python tools/isastat.py build/asm/api.s <substring of the mangled kernel name> [--all]
The main loop = the backward branch that spans the most MFMAs (the row-tile loop of the fused kernels); --all: first to last MFMA
of the kernel.  Prints the class counts, the VALU instructions by opcode and by what they are for (a reading of the opcodes:
activation, relu derivative / masks, packed fringe FMAs, AccVGPR traffic, moves / layout, address arithmetic, compares / selects)."""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))      # (not the first s_endpgm: a kernel may return early)
raw = lines[start:end]
body = [l.strip() for l in raw if l.strip() and not l.strip().startswith((";", ".section", ".p2align", ".type", ".globl"))]
if "--all" in sys.argv:
    mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
    lo, hi = mf[0], mf[-1] + 1
else:
    label = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: label[m.group(1)] = i
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.match(r"^s_cbranch\w*\s+(\.LBB\d+_\d+)", l) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label and label[m.group(1)] < i:
            n = sum(1 for k in range(label[m.group(1)], i) if body[k].startswith("v_mfma"))
            if n > best[0]: best = (n, label[m.group(1)], i + 1)
    lo, hi = best[1], best[2]
loop = [l for l in body[lo:hi] if not re.match(r"^\.LBB", l)]
cls = collections.Counter()
for l in loop:
    op = l.split()[0]
    if op.startswith("v_mfma"): cls["mfma " + op.split("_")[3]] += 1
    elif op.startswith("v_accvgpr"): cls["accvgpr mov"] += 1
    elif op.startswith("v_"): cls["valu"] += 1
    elif op.startswith("ds_read") or op.startswith("ds_load"): cls["lds read " + op] += 1
    elif op.startswith("ds_"): cls["lds write " + op] += 1
    elif op.startswith(("global_", "buffer_", "flat_")): cls["vmem " + op] += 1
    elif op == "s_waitcnt": cls["s_waitcnt"] += 1
    elif op == "s_nop": cls["s_nop"] += 1; cls["nop cycles"] += int(l.split()[1]) + 1
    elif op.startswith("s_"): cls["salu"] += 1
    else: cls["other " + op] += 1
print(f"instructions {lo}..{hi} of the kernel ({len(loop)})")
for k in sorted(cls): print(f"{k:32s} {cls[k]}")
valu = collections.Counter(l.split()[0] for l in loop if l.startswith("v_") and not l.startswith(("v_mfma", "v_accvgpr")))
print("VALU by opcode:", ", ".join(f"{k} {v}" for k, v in valu.most_common(30)))
CAT = [("activation (max / exp / rcp / tanh)", ("v_max_f32", "v_max_i32", "v_exp", "v_rcp", "v_log", "v_min_f32")),
       ("relu derivative / masks (mul_legacy, cndmask, cmp)", ("v_mul_legacy", "v_cndmask", "v_cmp")),
       ("packed f32 math (fringe dW rows, pair sums)", ("v_pk_fma", "v_pk_mul", "v_pk_add")),
       ("scalar f32 math", ("v_fma_f32", "v_fmac", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_mac")),
       ("moves / layout", ("v_mov", "v_perm", "v_swap", "v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_bfe", "v_and", "v_or")),
       ("address / integer arithmetic", ("v_add_u32", "v_add3", "v_lshl", "v_mul_lo", "v_mul_u32", "v_mad_u", "v_sub_u32", "v_ashr", "v_lshr", "v_add_co", "v_addc"))]
cat = collections.Counter()
for op, n in valu.items():
    for name, pre in CAT:
        if op.startswith(pre): cat[name] += n; break
    else: cat["other"] += n
for k, v in cat.most_common(): print(f"  {k:52s} {v}")
