"""hipcc with the MFMA hazard check INSIDE the compile (build.py and jit.py compile every kernel unit through `run`).

hipcc is asked for its own sub-commands (`-###` with `-save-temps`: preprocess, device bitcode, device LISTING, assemble, link, bundle, host side)
and they are replayed one by one in a scratch directory.  Between the step that writes the gfx950 listing and the step that assembles it,
`hazard_lint.fix_listing` reads the listing along its control flow and inserts the wait states (or `s_waitcnt`) every finding lacks -- so the
object that comes out is free of the pairs the check knows BY CONSTRUCTION, whatever the register allocator and the scheduler did around the
inline-asm MFMAs of this shape, and no unit is ever handed out unchecked: the check needs nothing but the compiler that is running anyway.
Afterwards, where llvm-objdump exists, the finished object is disassembled and checked once more (what the assembler made of it); a finding
there is an error.

`run` returns a Result: rc, stderr of every step (the -Rpass-analysis remarks the callers read), and `status` -- what the check did, in one line
(build.py links it into the library: tbnn_lint_status(); jit.py writes it next to the cached library)."""
import os
import shlex
import shutil
import subprocess
import tempfile

from . import hazard_lint


class Result:
    def __init__(self, rc, stderr, status="", fixed=(), left=()):
        self.rc, self.stderr, self.status, self.fixed, self.left = rc, stderr, status, list(fixed), list(left)


def _tally(found):
    t = {}
    for f in found:
        k = f[4].split()[0]
        t[k] = t.get(k, 0) + 1
    return " ".join(f"{k}:{v}" for k, v in sorted(t.items()))


def _plain(cmd, why: str) -> Result:
    cmd = [c for c in cmd if c != "-DTBNN_ASM_MFMA_NOP=0"]
    cmd = cmd[:1] + ["-DTBNN_ASM_MFMA_NOP=1"] + cmd[1:]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True)
    except OSError as e:
        return Result(-1, str(e))
    if p.returncode != 0:
        return Result(p.returncode, p.stderr)
    out = cmd[cmd.index("-o") + 1] if "-o" in cmd else None
    try:
        found = hazard_lint.check(out)
    except (OSError, subprocess.CalledProcessError, RuntimeError, ValueError, TypeError) as e:
        return Result(1, p.stderr + f"\n{cmd[-1]}:1:1: error: the unit cannot be checked for MFMA hazards (no compiler listing: {why.strip()[-200:]!r}; no disassembly: {e})\n")
    if found:
        return Result(1, p.stderr + f"\n{cmd[-1]}:1:1: error: MFMA hazards in a unit that could only be compiled plainly: {hazard_lint.describe(found)}\n", "", [], found)
    return Result(0, p.stderr, "compiled plainly with the wait states inside the asm MFMAs (the compiler driver printed no sub-commands); disassembly clean")


def run(cmd, keep_listing: str = None, verify: bool = True) -> Result:
    """`cmd`: a complete hipcc command line (list; the output after -o and the source with absolute paths), compiling ONE source to an object
    or a shared library.  `keep_listing`: copy the repaired device listing there (diagnostics)."""
    work = tempfile.mkdtemp(prefix="tbnn_cc_")
    err_all = []
    try:
        try:
            r = subprocess.run(list(cmd) + ["-save-temps", "-###"], cwd=work, capture_output=True, text=True)
        except OSError as e:
            return Result(-1, str(e))
        steps = [shlex.split(l) for l in r.stderr.splitlines() if l.startswith(' "')]
        is_listing = lambda st: "-S" in st and "-o" in st and "-triple" in st and st[st.index("-triple") + 1].startswith("amdgcn")
        if r.returncode != 0 or not any(is_listing(st) for st in steps):
            # a compiler driver that does not print its sub-commands this way: compile plainly with the wait states inside every asm MFMA and
            # let the disassembly decide -- clean, or no object
            return _plain(cmd, r.stderr)
        fixed, left, nlist = [], [], 0
        for st in steps:
            try:
                p = subprocess.run(st, cwd=work, capture_output=True, text=True)
            except OSError as e:
                return Result(-1, "\n".join(err_all + [str(e)]))
            err_all.append(p.stderr)
            if p.returncode != 0:
                return Result(p.returncode, "\n".join(err_all))
            if is_listing(st):
                out = st[st.index("-o") + 1]
                path = out if os.path.isabs(out) else os.path.join(work, out)
                text = open(path).read()
                new, fx, lf = hazard_lint.fix_listing(text)
                nlist += 1
                fixed += fx; left += lf
                if new is not text and fx:
                    with open(path, "w") as f:
                        f.write(new)
                if keep_listing:
                    shutil.copy(path, keep_listing)
        if nlist == 0:
            return Result(1, "\n".join(err_all) + "\nchecked_compile: hipcc produced no gfx950 listing to check\n")
        status = f"listing checked ({len(fixed)} repaired{': ' + _tally(fixed) if fixed else ''})"
        if hazard_lint.RULES_OFF:                                  # a diagnostic switch must not pass for a full check
            status += f" WITH RULES SWITCHED OFF: {','.join(sorted(hazard_lint.RULES_OFF))}"
        if left:
            return Result(1, "\n".join(err_all) + f"\n{cmd[-1]}:1:1: error: MFMA hazards without a local repair: {hazard_lint.describe(left)}\n", status, fixed, left)
        if verify and "-o" in cmd and hazard_lint.available():
            out = cmd[cmd.index("-o") + 1]
            try:
                again = hazard_lint.check(out)
            except (OSError, subprocess.CalledProcessError, RuntimeError, ValueError) as e:
                again = None
                status += f"; disassembly not checked ({type(e).__name__})"
            if again:
                return Result(1, "\n".join(err_all) + f"\n{cmd[-1]}:1:1: error: the assembled object still shows MFMA hazards: {hazard_lint.describe(again)}\n",
                              status, fixed, again)
            if again is not None:
                status += "; disassembly clean"
        return Result(0, "\n".join(err_all), status, fixed, left)
    finally:
        shutil.rmtree(work, ignore_errors=True)
