#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop from hipcc -S output.

    hipcc ... --cuda-device-only -S file.hip -o file.s
    python profiles/tools/isa_mix.py file.s <mangled-name-substring> [--all]

Finds the function whose label contains the substring, takes its longest basic-block run that ends in a backward
branch (the min-sum iteration body is one straight-line block of 24 unrolled rows) -- or the whole function with
--all -- and prints the count per issue class measured in profiles/r02_ubench_valu_issue_classes.txt:
  valu_full  : v_add/sub/mul/fma/fmac_f32 (any encoding without DPP/SDWA), v_and/or/xor_b32, v_add/sub_u32, v_mov_b32
  valu_half  : every other VALU instruction (min/max/med3, shifts, bfi, DPP, SDWA, compares, conversions ...)
  lds, salu, vmem, waitcnt, nop
"""
import json
import re
import sys

FULL = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mul_legacy)_f32(_e32|_e64)?$|^v_(and|or|xor)_b32(_e32|_e64)?$|"
                  r"^v_(add|sub|subrev)_u32(_e32|_e64)?$|^v_mov_b32(_e32|_e64)?$")


def classify(op):
    if op.startswith("v_"):
        return "valu_full" if FULL.match(op) else "valu_half"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def blocks(lines):
    cur_label, cur = None, []
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".")) and not t.startswith(".LBB"):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            yield cur_label, cur
            cur_label, cur = m.group(1), []
            continue
        if t.startswith(";;#"):
            continue
        op = t.split()[0]
        if re.match(r"^[a-z_0-9]+$", op):
            cur.append((op, t))
    yield cur_label, cur


def main():
    path, key = sys.argv[1], sys.argv[2]
    whole = "--all" in sys.argv
    text = open(path).read().splitlines()
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\S*:", l) and key in l)
    # (the function's end marker, not its first s_endpgm: the kernel has an early exit right behind its prologue)
    end = next(i for i in range(start + 1, len(text)) if text[i].strip().startswith(".Lfunc_end"))
    body = text[start + 1:end + 1]
    bl = list(blocks(body))
    if whole:
        ins = [x for _, b in bl for x in b]
    else:
        ins = max((b for _, b in bl), key=len)
    mix = {}
    for op, t in ins:
        c = classify(op)
        if c == "valu_full" and (" row_" in t or "quad_perm" in t or "_dpp" in op or "_sdwa" in op):
            c = "valu_half"
        mix[c] = mix.get(c, 0) + 1
    mix["total"] = len(ins)
    mix["dpp"] = sum(1 for op, t in ins if "_dpp" in op or "quad_perm" in t or " row_" in t)
    print(json.dumps(mix, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
