#!/usr/bin/env python3
"""Registers, LDS and scratch of the kernels in one object file of the library (code-object metadata).
    python tools/kernel_meta.py [msnap_aux.o] [name substring]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else "msnap_aux.o"
    if not os.path.isabs(obj):
        obj = os.path.join(ROOT, "drone_path_planning_python_amd", "csrc", obj)
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        co, fat = os.path.join(tmp, "k.co"), os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(tmp, "copy.o")], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
        name = name.split("(")[0].replace("void ", "")
        if pat in name:
            print(f"{name:60s} vgpr {get('vgpr_count'):>4s} agpr {blk.split()[1] if blk.split() else '?':>3s} sgpr {get('sgpr_count'):>4s} "
                  f"lds {get('group_segment_fixed_size'):>6s} scratch {get('private_segment_fixed_size'):>4s}")


if __name__ == "__main__":
    main()
