#!/usr/bin/env python3
"""Instruction classes of the innermost loop of a kernel (disassembly of the built code object), with the
issue-slot arithmetic DESIGN.md uses: 4-cycle class (v_mad_u64_u32, 64-bit shifts, v_mul_lo/hi, carry
adds) = 1 slot, other VALU = 0.55 slot (2.3 / 4.2 cycles, tools/ubench_valu.hip), s_nop = 0.25.
usage: isa_loop_table.py <object> <kernel name substring> [markdown title]"""
import collections
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_count import classify  # noqa: E402
from isa_hazard_check import LLVM, TARGET  # noqa: E402
import subprocess


def listing(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "f.bin"), os.path.join(tmp, "d.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"])
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else pat
    name, rows = None, []
    for line in listing(obj).splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            if name and pat in name:
                break
            name, rows = m.group(2), []
            continue
        m = re.match(r"^\t(.+?)\s*// ([0-9A-F]+):", line)
        if m and name:
            rows.append((int(m.group(2), 16), m.group(1).strip()))
    assert name and pat in name, "kernel not found"
    # innermost loop = the backward branch with the smallest span
    best = None
    for i, (addr, ins) in enumerate(rows):
        m = re.match(r"s_cbranch_\w+ (\d+)", ins)
        if not m:
            continue
        off = int(m.group(1))
        if off >= 0x8000:   # negative 16-bit word offset
            tgt = addr + 4 + (off - 0x10000) * 4
            j = next(k for k, (a, _) in enumerate(rows) if a == tgt)
            if best is None or i - j < best[1] - best[0]:
                best = (j, i)
    body = [ins for _, ins in rows[best[0]:best[1] + 1]]
    c = collections.Counter(classify(i.split()[0]) for i in body)
    mn = collections.Counter(i.split()[0] for i in body)
    four = c["mad64"] + c["carry"] + c["mul32"] + c["shift64"]
    slots = four + 0.55 * c["valu_other"] + 0.25 * c["s_nop"]
    print(f"### {title}\n")
    print(f"kernel `{name}`, innermost loop: {len(body)} instructions\n")
    print("| class | count | issue slots |\n|---|---|---|")
    for k, w in (("mad64", 1), ("mul32", 1), ("shift64", 1), ("carry", 1), ("valu_other", 0.55), ("s_nop", 0.25), ("salu", 0), ("vmem", 0), ("waitcnt", 0)):
        if c[k]:
            print(f"| {k} | {c[k]} | {c[k] * w:.1f} |")
    print(f"| **total** | {len(body)} | **{slots:.1f}** |\n")
    print("mnemonics:", ", ".join(f"{k} x{v}" for k, v in mn.most_common()))
    print()


if __name__ == "__main__":
    main()
