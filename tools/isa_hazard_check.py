#!/usr/bin/env python3
"""Build-time check of the gfx950 machine code for hazards the compiler cannot see.

The multipliers are inline asm (fp_mac_groups.inc, fp_mad_groups.inc, mac_1limb, the MFMA k-step):
LLVM's hazard recogniser does not look inside asm statements, so two properties that the kernels
rely on are re-checked on the disassembly of every translation unit after each build:

  1. VALU-written SGPR -> VALU read.  gfx940/gfx950 need two wait states between a VALU instruction
     that writes an SGPR or VCC (carry-outs of v_mad_u64_u32 / v_add*_co, v_cmp, v_readlane,
     v_readfirstlane) and a VALU instruction that reads it (carry-in, SGPR data operand).  An
     instruction in between is one wait state, `s_nop N` is N + 1.  A modulus limb materialised with
     v_readfirstlane right before the asm that multiplies by it would silently read a stale SGPR.
  3. After the last MFMA of those kernels no accumulator is read or copied before the MFMA's result latency has
     passed (the compiler may hoist its v_accvgpr_read above a bare `s_nop` asm: found on k_matmul_mfma_limbs<2>,
     where one output in sixteen came out wrong).
  2. No v_accvgpr_* between the first and the last MFMA of k_matmul_mfma_1limb / k_matmul_mfma_limbs: an accumulator copied
     between asm MFMA statements would get no MFMA-write hazard padding (and means the "+a" pinning
     of the accumulators failed).

usage: isa_hazard_check.py [object files ...]   (default: build/obj/fp_inst_*.o)
Exits non-zero and prints the offending instructions if a check fails."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
WAIT_STATES = int(os.environ.get("ISA_CHECK_WAIT_STATES", "2"))  # 2 on gfx940/gfx950 (override: self-test of this checker)

TWO_DST = re.compile(r"^v_(mad_u64_u32|mad_i64_i32|add_co_u32|sub_co_u32|subrev_co_u32|addc_co_u32|subb_co_u32|subbrev_co_u32|"
                     r"div_scale_f32|div_scale_f64)")
SGPR_DST = re.compile(r"^v_(cmp_|cmpx_|readlane_b32|readfirstlane_b32)")


def disassemble(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}",
                           f"--output={co}"])
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)


def functions(listing):
    name, body = None, []
    for line in listing.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name and line.startswith("\t"):
            ins = line.split("//")[0].strip()
            if ins:
                body.append(ins)
    if name:
        yield name, body


def sregs(op):
    """SGPR indices named by one operand (vcc = 106, 107)."""
    op = op.strip()
    if op in ("vcc", ):
        return {106, 107}
    if op == "vcc_lo":
        return {106}
    if op == "vcc_hi":
        return {107}
    m = re.fullmatch(r"s(\d+)", op)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def split(ins):
    parts = ins.split(None, 1)
    mnem = parts[0]
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    # drop modifiers such as "clamp", "op_sel:[..]"
    ops = [o.split(" ")[0] for o in ops]
    return mnem, ops


def valu_sgpr_defs_uses(ins):
    mnem, ops = split(ins)
    if not mnem.startswith("v_"):
        return mnem, set(), set()
    ndst = 2 if TWO_DST.match(mnem) else 1
    defs, uses = set(), set()
    for i, o in enumerate(ops):
        (defs if i < ndst else uses).update(sregs(o))
    if SGPR_DST.match(mnem) is None and not TWO_DST.match(mnem):
        defs = set()          # a plain VALU op has a VGPR destination
    # e32 carry ops read vcc implicitly only when it is printed; objdump prints it, nothing to add
    return mnem, defs, uses


def check_sgpr_hazard(name, body):
    bad = []
    recent = []   # (wait states since, defs, text)
    for ins in body:
        mnem, defs, uses = valu_sgpr_defs_uses(ins)
        if mnem.startswith("v_") and uses:
            for age, d, text in recent:
                if age < WAIT_STATES and d & uses:
                    bad.append((name, text, ins, age))
        step = 1
        if mnem == "s_nop":
            step = int(ins.split()[1], 0) + 1
        recent = [(age + step, d, t) for age, d, t in recent if age + step < WAIT_STATES]
        if defs:
            recent.append((0, defs, ins))
    return bad


def mfma_accumulators(body):
    """AGPR numbers that some MFMA of the function writes (the accumulators; the compiler may also park spilled VGPRs in
    the AGPRs left over, and moving those is no hazard)."""
    regs = set()
    for ins in body:
        if ins.startswith("v_mfma"):
            m = re.match(r"v_mfma\S*\s+a\[(\d+):(\d+)\]", ins)
            if m:
                regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def touches(ins, regs):
    return any(int(r) in regs for r in re.findall(r"\ba(\d+)\b", ins))


def check_mfma_region(name, body):
    idx = [i for i, ins in enumerate(body) if ins.startswith("v_mfma")]
    if not idx:
        return []
    regs = mfma_accumulators(body)
    return [(name, ins) for ins in body[idx[0]:idx[-1] + 1] if ins.startswith("v_accvgpr") and touches(ins, regs)]


MFMA_READ_WAIT = 18   # wait states between an MFMA's issue and a VALU read of its result registers (8-pass int8 MFMA: 18)


def check_mfma_tail(name, body):
    """The asm MFMAs' writes are invisible to the hazard recogniser: after the LAST one, an accumulator must not be read or
    copied (v_accvgpr_read / v_accvgpr_mov) before MFMA_READ_WAIT wait states have passed.  Scans forward in address
    order, which is execution order up to the first unconditional branch (where the scan stops)."""
    idx = [i for i, ins in enumerate(body) if ins.startswith("v_mfma")]
    if not idx:
        return []
    waited = 0
    regs = mfma_accumulators(body)
    for ins in body[idx[-1] + 1:]:
        mnem = ins.split()[0]
        if mnem in ("s_branch", "s_endpgm", "s_setpc_b64"):
            break
        if mnem in ("v_accvgpr_read_b32", "v_accvgpr_mov_b32") and waited < MFMA_READ_WAIT and touches(ins, regs):
            return [(name, ins, waited)]
        waited += int(ins.split()[1], 0) + 1 if mnem == "s_nop" else 1
        if waited >= MFMA_READ_WAIT:
            break
    return []


def main(argv):
    objs = argv[1:] or sorted(glob.glob(os.path.join(ROOT, "build", "obj", "fp_inst_*.o")))
    objs = [o for o in objs if re.search(r"fp_inst_\d+\.o$", o) or argv[1:]]
    if not objs:
        print("isa_hazard_check: no objects (build the library first)")
        return 2
    total_fn = total_ins = 0
    hazards, acc, tail = [], [], []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            for name, body in functions(disassemble(obj, tmp)):
                total_fn += 1
                total_ins += len(body)
                hazards += check_sgpr_hazard(name, body)
                if "k_matmul_mfma" in name:
                    acc += check_mfma_region(name, body)
                    tail += check_mfma_tail(name, body)
    for name, w, r, age in hazards[:40]:
        print(f"HAZARD {name}: `{w}` then `{r}` after {age} wait state(s)")
    for name, ins in acc[:40]:
        print(f"ACCVGPR inside the MFMA stream of {name}: `{ins}`")
    for name, ins, waited in tail[:40]:
        print(f"MFMA RESULT READ TOO EARLY in {name}: `{ins}` {waited} wait state(s) after the last MFMA")
    print(f"isa_hazard_check: {len(objs)} objects, {total_fn} kernels, {total_ins} instructions, "
          f"{len(hazards)} SGPR hazards, {len(acc)} accvgpr moves inside the MFMA stream, "
          f"{len(tail)} accumulator reads within {MFMA_READ_WAIT} wait states of the last MFMA")
    return 1 if hazards or acc or tail else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
