#!/usr/bin/env python3
"""tools/salu_audit.py — what the scalar instructions of a decode kernel ARE (520 k SALU against 799 k VALU
instructions per wave-decode of the list-of-32 kernel "with every shape constant" looked suspicious). Runs HERE (no GPU):
extracts the gfx950 code object of a translation unit (llvm-objdump --offloading), disassembles one kernel and classifies
every instruction — whole kernel, and per innermost loop (backward branch) for the largest loops.

    python tools/salu_audit.py [object=polar_amd/_build/polar_kernels.hip.ed32.o] [kernel-substring=ILi32ELi3ELi0ELb1ELi11ELi0E]

Classes of scalar instructions:
  literal     s_mov_b32 / s_mov_b64 of a constant: fp64 literals (a VOP3 instruction cannot carry a 64-bit literal: every polynomial
              coefficient, threshold and mask that is not an inline constant is built in an SGPR pair next to its use) and masks
  mask        exec / vcc / lane-mask logic: s_and_b64, s_or_b64, s_andn2_b64, s_xor_b64, s_*_saveexec, s_cselect_b64, s_not, s_bcnt, s_ff1
              (divergence handling, the wave-mask form of the rare-path tests, ballots ranked on the scalar unit)
  branch      s_cbranch_*, s_branch, s_setpc, s_call
  compare     s_cmp_*, s_bitcmp
  address     integer / address arithmetic: s_add, s_addc, s_sub, s_lshl, s_lshr, s_mul, s_and_b32, s_or_b32, s_bfe, s_cselect_b32, s_min/max
  move        s_mov of a register
  smem        s_load_*, s_buffer_load_*
  wait        s_waitcnt, s_nop (not ALU work)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(obj))
        subprocess.check_call(["cp", obj, tmp])
        subprocess.check_call([OBJDUMP, "--offloading", tmp], stdout=subprocess.DEVNULL, cwd=d)
        co = [f for f in os.listdir(d) if "amdgcn" in f]
        assert co, "no device code object in " + obj
        return subprocess.check_output([OBJDUMP, "-d", os.path.join(d, co[0])], text=True)


def klass(op, args):
    if op in ("s_waitcnt", "s_nop", "s_waitcnt_depctr", "s_sleep"):
        return "wait"
    if not op.startswith("s_"):
        return "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call", "s_swappc", "s_endpgm", "s_barrier", "s_getpc")):
        return "branch"
    if op.startswith(("s_cmp", "s_bitcmp")):
        return "compare"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op in ("s_mov_b32", "s_mov_b64", "s_movk_i32"):
        src = args.split(",")[-1].strip()
        return "move" if re.match(r"^(s\d|s\[|vcc|exec|ttmp|m0)", src) else "literal"
    if op.endswith("_b64") and op.startswith(("s_and", "s_or", "s_xor", "s_andn2", "s_orn2", "s_nand", "s_nor", "s_xnor", "s_not", "s_cselect", "s_bcnt", "s_ff", "s_flbit", "s_wqm")) \
            or "saveexec" in op or op.startswith(("s_bcnt", "s_ff1", "s_ff0", "s_flbit")):
        return "mask"
    return "address"


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "polar_amd", "_build", "polar_kernels.hip.ed32.o")
    want = sys.argv[2] if len(sys.argv) > 2 else "ILi32ELi3ELi0ELb1ELi11ELi0E"
    text = disassemble(obj)
    m = re.search(r"^[0-9a-f]+ <(\S*%s\S*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)" % re.escape(want), text, flags=re.S | re.M)
    assert m, "kernel not found"
    name, body = m.group(1), m.group(2)
    ins = []
    for l in body.split("\n"):
        q = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
        if q:
            ins.append((int(q.group(3), 16), q.group(1), q.group(2)))
    idx = {a: i for i, (a, _, _) in enumerate(ins)}
    print("kernel", name)
    print("object", os.path.relpath(obj, ROOT), "- %d instructions, %.1f KiB of code" % (len(ins), (ins[-1][0] - ins[0][0]) / 1024.0))

    def mix(s, e):
        c = collections.Counter()
        for a, op, args in ins[s:e + 1]:
            c[klass(op, args)] += 1
        return c

    def show(c):
        salu = sum(c[k] for k in ("literal", "mask", "branch", "compare", "address", "move", "smem"))
        return "VALU %5d  SALU %5d (literal %d, mask %d, branch %d, compare %d, address %d, move %d, smem %d)  wait/nop %d  LDS %d  VMEM %d" % (
            c["valu"], salu, c["literal"], c["mask"], c["branch"], c["compare"], c["address"], c["move"], c["smem"], c["wait"], c["lds"], c["vmem"])
    print("whole kernel (static):", show(mix(0, len(ins) - 1)))
    lit = collections.Counter()
    for a, op, args in ins:
        if klass(op, args) == "literal":
            lit[args.split(",")[-1].strip()] += 1
    print("most frequent literals (halves of fp64 constants and masks):", ", ".join("%s x%d" % kv for kv in lit.most_common(14)))
    loops = set()
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            q = re.match(r"(-?\d+)", args)
            if q:
                off = int(q.group(1))
                off -= 65536 if off >= 32768 else 0
                t = a + 4 + off * 4
                if t < a and t in idx:
                    loops.add((idx[t], i))
    inner = [(s, e) for s, e in loops if not any(s <= s2 and e2 <= e and (s2, e2) != (s, e) for s2, e2 in loops)]
    print("%d loops, %d innermost; the 14 largest innermost loops (address: instructions):" % (len(loops), len(inner)))
    for s, e in sorted(inner, key=lambda x: -(x[1] - x[0]))[:14]:
        c = mix(s, e)
        rcp = sum(1 for a, op, args in ins[s:e + 1] if op.startswith("v_rcp_f64"))
        print("  0x%06x: %4d  %s  [%d divisions]" % (ins[s][0], e - s + 1, show(c), rcp))


if __name__ == "__main__":
    main()
