#!/usr/bin/env python3
"""Static check of the gfx9 DPP hazard in AMDGPU assembly listings (hipcc -S): a VGPR read through a DPP lane permutation must
not have been written by the previous two wait states (VALU instructions; `s_nop N` counts N + 1).  The compiler pads its own DPP
instructions; the hand-written blocks of csrc/mppi_quad.hpp / mppi_oct.hpp (rotations folded into v_fmac_f32_dpp, row_ror:8
exchanges) are invisible to its hazard recogniser, so their spacing is by construction - and verified here, over every instruction
of the listing, whoever emitted it.  At a label the write history is unknown (control flow may arrive from a block that has just
written the operand): the compiler's own DPP instructions are trusted there, a HAND-WRITTEN DPP form within two wait states of a
label is reported.
Second hazard checked the same way (gfx940+, "trans forwarding"): a non-transcendental VALU instruction must not read the result
of a transcendental one (v_sin / v_cos / v_rcp / v_rsq / v_sqrt / v_exp / v_log) in the very next issue slot - again padded by the
compiler for its own code only (an inline-assembly block that took (cos q, sin q) right behind v_sin_f32 rotated with a stale sine).
    python tools/check_dpp_hazards.py file.s [...]      -> exit status 1 and the offending lines if a hazard is found
    python tools/check_dpp_hazards.py --library libmppi_hip.so   -> the same over the disassembly of every gfx950 code object in the
                                                                 library (llvm-objdump --offloading, then -d)"""
import re
import sys

TRANS = re.compile(r"^v_(sin|cos|rcp|rcp_iflag|rsq|sqrt|exp|log|exp_legacy|log_legacy)_(f32|f16|bf16)")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def hand_written(op, code):
    """DPP forms only the inline-assembly blocks emit (the compiler never forms v_fmac_f32_dpp - csrc/mppi_quad.hpp - nor row_ror:8)"""
    return op == "v_fmac_f32_dpp" or "row_ror:8" in code


def check(path):
    bad = 0
    recent = []   # [(wait states ago, set of vgprs written)]
    trans = set()  # vgprs written by a transcendental in the previous issue slot
    since_label = 99  # wait states since the last label (control flow may arrive from anywhere)
    for ln, line in enumerate(open(path, errors="replace"), 1):
        code = line.split(";")[0].split("//")[0].strip()   # (hipcc -S comments with ';', llvm-objdump -d with '//')
        if not code or code.endswith(":") or code.startswith("."):
            if code.endswith(":"):
                trans = set()
                recent = []      # a label: control flow may come from anywhere - the compiler's own padding is trusted across blocks ...
                since_label = 0  # ... but not for the hand-written DPP forms: the compiler does not know they read through DPP
            continue
        parts = code.split(None, 1)
        op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        if op == "s_nop":
            trans = set()
            n = int(rest.strip(), 0) + 1
            recent = [(a + n, w) for a, w in recent if a + n <= 2]
            since_label += n
            continue
        ops = [o.strip() for o in rest.split(",")]
        if "_dpp" in op and len(ops) >= 2 and since_label < 2 and hand_written(op, code):
            print(f"{path}:{ln}: hand-written DPP read {since_label} wait state(s) behind a label (a predecessor block may just have written its operand): {code}")
            bad += 1
        if "_dpp" in op and len(ops) >= 2:
            src0 = ops[1].split()[0].lstrip("-|")
            for r in regs(src0):
                for ago, written in recent:
                    if ago < 2 and r in written:
                        print(f"{path}:{ln}: DPP read of v{r} {ago} wait state(s) after its write: {code}")
                        bad += 1
        is_valu = op.startswith("v_")
        if is_valu and trans and not TRANS.match(op):
            for src in ops[1:]:
                hit = regs(src.split()[0]) & trans if src else set()
                if hit:
                    print(f"{path}:{ln}: v{sorted(hit)[0]} read in the issue slot after the transcendental that wrote it: {code}")
                    bad += 1
        trans = regs(ops[0]) if (is_valu and TRANS.match(op) and ops) else set()
        since_label += 1
        if is_valu or not op.startswith("s_"):   # every non-scalar instruction advances the wait states
            recent = [(a + 1, w) for a, w in recent if a + 1 <= 2]
        elif op.startswith("s_"):
            recent = [(a + 1, w) for a, w in recent if a + 1 <= 2]
        if is_valu and ops and not op.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
            recent.append((0, regs(ops[0])))
    return bad


def check_library(lib, objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
    """disassemble every gfx950 code object of a HIP shared library and check it; returns (hazards, dpp instructions seen)"""
    import glob
    import os
    import shutil
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="dpp_check_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
        bad = seen = 0
        for co in sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950"))):
            dis = co + ".dis"
            with open(dis, "w") as f:
                subprocess.run([objdump, "-d", co], stdout=f, check=True)
            seen += sum(1 for line in open(dis, errors="replace") if "_dpp" in line)
            bad += check(dis)
        return bad, seen
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--library":
        total, seen = check_library(sys.argv[2])
        print(f"{total} DPP hazard(s) among {seen} DPP instructions of {sys.argv[2]}")
    else:
        total = sum(check(p) for p in sys.argv[1:])
        print(f"{total} DPP hazard(s) in {len(sys.argv) - 1} file(s)")
    sys.exit(1 if total else 0)
