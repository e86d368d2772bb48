#!/usr/bin/env python3
"""Build-time check of the inline-assembly granule loads (grid_sync.hpp / solver_pipe.hpp load_pair).

`load_pair` issues `global_load_dwordx4 vDST, vADDR, off sc1` from inline asm; the compiler's own wait counting does
not see it, so the code relies on every such load being followed by an explicit `s_waitcnt vmcnt(0)` BEFORE any
instruction reads or overwrites vDST.  That holds by construction of the source and by register allocation; this
script verifies it in the ISA actually built: it extracts the gfx950 code object from the object file / library,
disassembles it and walks every such load forward to the next `s_waitcnt` whose vmcnt is 0 (following the code layout through
conditional branches; an unconditional branch or the end of the program first is reported), failing if any instruction in between names one of the four destination registers.

    python tools/check_async_loads.py [rdis_amd/lib/obj/rdis_hip.o]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout


def regs_of(tok):
    """the vector registers an operand token names: v12 -> {12}, v[4:7] -> {4, 5, 6, 7}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check(text):
    loads, bad, open_ended = 0, [], 0
    func = "?"
    lines = text.splitlines()
    insn = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//")
    for i, ln in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            func = m.group(1)
            continue
        m = insn.match(ln)
        if not m or m.group(1) != "global_load_dwordx4" or " sc1" not in " " + m.group(2):
            continue
        ops = [t.strip() for t in m.group(2).split(",")]
        dst = regs_of(ops[0])
        if len(dst) != 4:
            continue
        loads += 1
        for j in range(i + 1, min(i + 4000, len(lines))):
            mj = insn.match(lines[j])
            if not mj:
                continue    # (labels, blank lines)
            op, rest = mj.group(1), mj.group(2)
            if op == "s_waitcnt" and re.search(r"vmcnt\(0\)", rest):
                break
            if op == "s_branch" or op == "s_endpgm" or op == "s_setpc_b64":
                open_ended += 1
                break
            # (a conditional branch: the loads sit in `if (still missing) load` blocks laid out inline -- the scan follows the
            # layout through them, which is where a copy at the join of the conditional would stand)
            if op.startswith("global_load") and " sc1" in " " + rest:
                continue    # the next load of the batch (its own destination is checked on its own turn)
            used = set()
            for t in re.findall(r"v\[\d+:\d+\]|v\d+", rest):
                used |= regs_of(t)
            if used & dst:
                bad.append((func, lines[i].strip(), lines[j].strip()))
                break
    return loads, bad, open_ended


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rdis_amd", "lib", "obj", "rdis_hip.o")
    loads, bad, open_ended = check(disassemble(obj))
    print("%d asynchronous granule loads checked, %d touched before their s_waitcnt vmcnt(0), %d reach a branch first" % (loads, len(bad), open_ended))
    for f, a, b in bad[:10]:
        print("  in %s:\n     %s\n     %s" % (f, a, b))
    return 1 if bad or loads == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
