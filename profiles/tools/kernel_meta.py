#!/usr/bin/env python3
"""Register / spill / LDS metadata of the kernels inside a host object or shared library built by hipcc.

    python profiles/tools/kernel_meta.py channelcoding_amd/csrc/build/geo_g255_24_p0.o [name-filter]

The device code object is the `hipv4-amdgcn-amd-amdhsa--gfx950` bundle of the `.hip_fatbin` section; its
NT_AMDGPU_METADATA note lists vgpr_count, vgpr_spill_count, private_segment_fixed_size ... per kernel.
Used by tests/test_host_logic.py (no dispatched instantiation may spill) and for DESIGN.md's register figures.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_meta(path):
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "fat.bin"), os.path.join(t, "dev.co")
        r = subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path], capture_output=True)
        if r.returncode != 0:  # an object without device code
            return []
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    out = []
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            out.append(cur)
            cur = {}
        if k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                 "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = int(v)
        elif k == "name" and not v.startswith("'") and "kernel" in v or (k == "name" and v.startswith("_Z")):
            cur["name"] = v
    if cur.get("name"):
        out.append(cur)
    for k in out:
        k["demangled"] = demangle(k["name"])
    return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernel_meta(sys.argv[1]):
        if flt in k["demangled"]:
            print("%-110s vgpr %3d agpr %3d spill %3d scratch %4d B lds %6d" % (
                k["demangled"][:110], k.get("vgpr_count", -1), k.get("agpr_count", -1), k.get("vgpr_spill_count", -1),
                k.get("private_segment_fixed_size", -1), k.get("group_segment_fixed_size", -1)))
