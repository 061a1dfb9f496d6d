#!/usr/bin/env python3
"""Loops of a gfx950 kernel from its disassembly: every backward branch with the instruction count (by class) of the range it
closes - the step / substep loops of the rollout kernels show up with their issue-slot counts.
    python tools/loop_stats.py <object or library> <kernel name filter> [min instructions]"""
import glob, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"

def main():
    path, filt = sys.argv[1], sys.argv[2]
    least = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    tmp = tempfile.mkdtemp(prefix="loops_")
    try:
        shutil.copy(path, os.path.join(tmp, "in.bin"))
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "in.bin"], cwd=tmp, check=True, capture_output=True)
        for co in sorted(glob.glob(os.path.join(tmp, "in.bin.*gfx950*"))):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", co], capture_output=True, text=True, check=True).stdout
            cur, ins = None, {}
            for line in dis.splitlines():
                m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
                if m:
                    cur = m.group(2)
                    ins[cur] = []
                    continue
                m = re.match(r"^\s+(\S.*?)\s*// ([0-9A-F]+): ([0-9A-F]{8})", line)
                if cur is not None and m:
                    ins[cur].append((int(m.group(2), 16), m.group(1).strip(), int(m.group(3), 16)))
            for name, lst in ins.items():
                if filt not in name:
                    continue
                print(name[:150])
                addr_index = {a: i for i, (a, _, _) in enumerate(lst)}
                for i, (a, code, word) in enumerate(lst):
                    op = code.split()[0]
                    if op.startswith(("s_cbranch", "s_branch")):
                        simm = word & 0xFFFF
                        if simm & 0x8000:
                            simm -= 0x10000
                        target = a + 4 + 4 * simm
                        if target <= a and target in addr_index:
                            body = lst[addr_index[target]:i + 1]
                            if len(body) < least:
                                continue
                            cls = {}
                            for _, c, _ in body:
                                o = c.split()[0]
                                k = ("nop" if o.startswith("s_nop") else "wait" if o.startswith("s_waitcnt") else "branch" if o.startswith(("s_cbranch", "s_branch")) else
                                     "salu" if o.startswith("s_") else "lds" if o.startswith("ds_") else "vmem" if o.startswith(("global_", "buffer_", "flat_", "scratch_")) else
                                     "acc" if o.startswith("v_accvgpr") else "valu" if o.startswith("v_") else "other")
                                cls[k] = cls.get(k, 0) + 1
                                if "_dpp" in o:
                                    cls["dpp"] = cls.get("dpp", 0) + 1
                                if o in ("v_mov_b32_e32", "v_mov_b32"):
                                    cls["mov"] = cls.get("mov", 0) + 1
                            print(f"  loop of {len(body):5d} instructions at +{target - lst[0][0]:#x}: " + " ".join(f"{k}={v}" for k, v in sorted(cls.items())))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

if __name__ == "__main__":
    main()
