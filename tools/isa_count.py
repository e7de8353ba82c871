#!/usr/bin/env python3
"""Instruction mix of kernels in a built translation unit (disassembly of the gfx950 code object).
usage: isa_count.py <object file> <substring of the mangled kernel name> [...]
Counts are static (whole kernel body); the 4-cycle class is what bounds the VALU-heavy kernels
(v_mad_u64_u32, v_addc/v_add_co/v_subb, v_mul_lo/hi, 64-bit shifts)."""
import collections
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_hazard_check import disassemble, functions  # noqa: E402


def classify(m):
    if m.startswith("v_mad_u64_u32") or m.startswith("v_mad_i64_i32"):
        return "mad64"
    if m.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co", "v_add_co", "v_sub_co", "v_subrev_co")):
        return "carry"
    if m.startswith(("v_mul_lo_u32", "v_mul_hi_u32")):
        return "mul32"
    if m.startswith(("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_lshl_add_u64")):
        return "shift64"
    if m.startswith("v_mfma"):
        return "mfma"
    if m.startswith("v_"):
        return "valu_other"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("s_nop"):
        return "s_nop"
    if m.startswith("s_waitcnt"):
        return "waitcnt"
    return "salu"


def main():
    obj, pats = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        for name, body in functions(disassemble(obj, tmp)):
            if pats and not any(p in name for p in pats):
                continue
            c = collections.Counter(classify(i.split()[0]) for i in body)
            other = collections.Counter(i.split()[0] for i in body if classify(i.split()[0]) == "valu_other")
            four = c["mad64"] + c["carry"] + c["mul32"] + c["shift64"]
            print(name)
            print("   total %d | 4-cycle VALU %d (mad64 %d carry %d mul32 %d shift64 %d) | other VALU %d | vmem %d lds %d salu %d s_nop %d waitcnt %d"
                  % (len(body), four, c["mad64"], c["carry"], c["mul32"], c["shift64"], c["valu_other"], c["vmem"], c["lds"],
                     c["salu"], c["s_nop"], c["waitcnt"]))
            print("   other VALU:", dict(other.most_common(12)))


if __name__ == "__main__":
    main()
