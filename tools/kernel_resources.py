#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in libgripnet_hip.so, read from the gfx950 code objects' metadata notes
(no GPU needed).  `python tools/kernel_resources.py [pattern]`; tests/test_abi.py asserts the budgets of the kernels whose
occupancy the design depends on."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".private_segment_fixed_size", ".vgpr_spill_count",
          ".sgpr_spill_count", ".group_segment_fixed_size", ".max_flat_workgroup_size")


def kernel_resources(lib=None):
    """{demangled kernel name: {field: int}} of every kernel in the library's gfx950 code objects."""
    lib = lib or os.path.join(ROOT, "gripnet_amd", "lib", "libgripnet_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)                       # llvm-objdump writes the bundles next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True)
        for name in sorted(os.listdir(tmp)):
            if "gfx950" not in name:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, name)],
                                   check=True, capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                block = ".agpr_count:" + block
                m = re.search(r"\.name:\s+(\S+)", block)
                if not m:
                    continue
                mangled = m.group(1).strip("'\"")
                res = {}
                for f in FIELDS:
                    v = re.search(re.escape(f) + r":\s+(\d+)", block)
                    if v:
                        res[f] = int(v.group(1))
                out[mangled] = res
    if out:
        names = list(out)
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
        out = {d.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""): out[n] for n, d in zip(names, dem)}
    return out


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for k, v in sorted(kernel_resources().items()):
        if pat in k and "rocprim" not in k:
            print("{:60s} vgpr {:3d} agpr {:3d} sgpr {:3d} scratch {:4d} spills {}/{} lds {}".format(
                k[:60], v.get(".vgpr_count", -1), v.get(".agpr_count", -1), v.get(".sgpr_count", -1),
                v.get(".private_segment_fixed_size", -1), v.get(".vgpr_spill_count", -1), v.get(".sgpr_spill_count", -1),
                v.get(".group_segment_fixed_size", -1)))
