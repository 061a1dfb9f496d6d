#!/usr/bin/env python3
"""Per-kernel resource table of a built HIP object / library: VGPRs, AGPRs, SGPRs, scratch bytes, static LDS, code bytes and -
from the disassembly - instruction counts by class (VALU / SALU / LDS / VMEM / branches / s_nop / s_waitcnt) of the whole
kernel.  Reads the AMDGPU metadata note (msgpack) of every gfx950 code object (llvm-objdump --offloading + llvm-readelf).
    python tools/kernel_stats.py mppi-isaac_amd/csrc/libmppi_hip.so [name filter ...]"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path, tmp):
    shutil.copy(path, os.path.join(tmp, "in.bin"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "in.bin"], cwd=tmp, check=True, capture_output=True)
    return sorted(glob.glob(os.path.join(tmp, "in.bin.*gfx950*")))


def metadata(co):
    out = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = [], None
    for line in out.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:
            cur = {"agpr": int(m.group(1))}
            kernels.append(cur)
            continue
        if cur is None:
            continue
        for key, tag in ((".vgpr_count", "vgpr"), (".sgpr_count", "sgpr"), (".private_segment_fixed_size", "scratch"), (".group_segment_fixed_size", "lds"),
                         (".vgpr_spill_count", "vspill"), (".sgpr_spill_count", "sspill")):
            m = re.match(r"\s+%s:\s+(\d+)" % re.escape(key), line)
            if m:
                cur[tag] = int(m.group(1))
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            cur["name"] = m.group(1)
    return [k for k in kernels if "name" in k]


def classify(op):
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "vmem"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("v_"):
        return "valu"
    return "other"


def instruction_counts(co):
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
    counts, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = counts.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        code = line.split("//")[0].strip()
        if not code or code.endswith(":"):
            continue
        op = code.split()[0]
        c = classify(op)
        cur[c] = cur.get(c, 0) + 1
        cur["total"] = cur.get("total", 0) + 1
        if "_dpp" in op:
            cur["dpp"] = cur.get("dpp", 0) + 1
        if op.startswith("v_pk_"):
            cur["pk"] = cur.get("pk", 0) + 1
    return counts


def demangle(name):
    for tool in (f"{LLVM}/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip()
            if out and out != name:
                return out
        except OSError:
            pass
    return name


def main():
    path, filters = sys.argv[1], sys.argv[2:]
    tmp = tempfile.mkdtemp(prefix="kstats_")
    try:
        rows = []
        for co in code_objects(path, tmp):
            ic = instruction_counts(co)
            for k in metadata(co):
                d = demangle(k["name"])
                short = re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "")).replace("void ", "").replace("mppi::", "")
                if filters and not any(f in short for f in filters):
                    continue
                c = ic.get(k["name"], {})
                if not any(r[0] == short and r[1] == k for r in rows):   # (non-template kernels repeat in every unit)
                    rows.append((short, k, c))
        print(f"{'kernel':<78} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'scr B':>6} {'lds B':>6} {'insts':>6} {'valu':>6} {'salu':>5} {'lds':>5} {'vmem':>5} {'scr':>4} {'acc':>4} {'br':>4} {'nop':>4} {'wait':>4} {'dpp':>5} {'pk':>4}")
        for short, k, c in sorted(rows, key=lambda r: r[0]):
            print(f"{short[:78]:<78} {k.get('vgpr', 0):>4} {k.get('agpr', 0):>4} {k.get('sgpr', 0):>4} {k.get('scratch', 0):>6} {k.get('lds', 0):>6} {c.get('total', 0):>6} "
                  f"{c.get('valu', 0):>6} {c.get('salu', 0):>5} {c.get('lds', 0):>5} {c.get('vmem', 0):>5} {c.get('scratch', 0):>4} {c.get('acc', 0):>4} {c.get('branch', 0):>4} "
                  f"{c.get('nop', 0):>4} {c.get('wait', 0):>4} {c.get('dpp', 0):>5} {c.get('pk', 0):>4}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
