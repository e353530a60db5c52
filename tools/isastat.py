"""dev: instruction mix of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S):
python tools/isastat.py build/asm/mid.s <substring of the mangled kernel name>
counts between the first and the last v_mfma of the kernel (the row loop + whatever sits between)"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = [l.strip() for l in lines[start:end] if l.strip() and not l.strip().startswith((";", "."))]
mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
loop = body[mf[0]:mf[-1] + 1]
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
for k in sorted(cls): print(f"{k:32s} {cls[k]}")
valu = collections.Counter(l.split()[0] for l in loop if l.startswith("v_") and not l.startswith(("v_mfma", "v_accvgpr")))
print("VALU by opcode:", ", ".join(f"{k} {v}" for k, v in valu.most_common(25)))
