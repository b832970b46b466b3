#!/usr/bin/env python3
"""Build-time check of the hand-counted cross-tile prefetch of solve_kernel_reg (csrc/msnap_solve.hip).

The persistent solve issues the next tile's input loads from inline asm during the last two
segments of a tile and retires them with `s_waitcnt vmcnt(N)`, N = the store instructions issued
after them (the compiler does not see asm loads, so nothing else would wait for them, and a wait
that is too loose would read registers whose loads have not landed).  This script disassembles the
gfx950 code object inside msnap_solve.o and checks, for every solve_kernel_reg instance:
  1. the in-loop prefetch burst exists (UW dwordx4 + UT dwordx2 loads back to back);
  2. the kernel holds exactly MAXM x kStoresPerSeg coefficient stores (global_store_dwordx4: 4 per
     segment at order 7, 5 at order 9) -- the compiler neither merged, split nor dropped one, so the
     two segments that follow the prefetch in the source issue 2 x kStoresPerSeg of them -- and an
     `s_waitcnt vmcnt(N)` with that N (launches with n_seg >= 2) and one with N / 2 (n_seg == 1);
  3. where the code after the burst is laid out contiguously up to the end of the kernel (the order-7
     instances), exactly 2 x kStoresPerSeg stores follow it and none of those instructions reads a
     destination register of the burst.
Exit status 1 (with a message) on a violation.   python3 tools/check_prefetch_isa.py [msnap_solve.o]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "solve.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(tmp, "copy.o")],
                       check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}",
                        f"--input={fat}", f"--output={co}"], check=True)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True,
                              capture_output=True, text=True).stdout


def regs(tok):
    """v[a:b] / vN operands of one instruction's operand string -> set of VGPR numbers."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def check_kernel(name, body):
    k = int(re.search(r"ILi(\d+)ELi(\d+)E", name).group(1))          # K = 4 (order 7) or 5 (order 9)
    stores_per_seg = 4 if k == 4 else 5
    ins = [ln.split(None, 1) for ln in body if ln and not ln.endswith(":")]
    ins = [(p[0], p[1] if len(p) > 1 else "") for p in ins]
    # prefetch bursts: maximal runs of global_load_dwordx4 / x2 with only address arithmetic between them
    bursts, i = [], 0
    while i < len(ins):
        if ins[i][0] == "global_load_dwordx4":
            j, loads = i, []
            while j < len(ins) and (ins[j][0].startswith("global_load_dwordx") or ins[j][0].startswith(("v_", "s_"))) \
                    and not ins[j][0].startswith(("s_waitcnt", "s_cbranch", "s_barrier")) and j - i < 200:
                if ins[j][0].startswith("global_load_dwordx"):
                    loads.append(j)
                j += 1
            if len(loads) >= 4:
                bursts.append(loads)
                i = loads[-1] + 1
                continue
        i += 1
    if len(bursts) < 2:
        return f"{name}: expected the prologue and the in-loop prefetch bursts, found {len(bursts)}"
    loads = bursts[-1]                      # the in-loop one comes last in program order
    dest = set()
    for j in loads:
        dest |= regs(ins[j][1].split(",")[0])
    maxm = int(re.search(r"ILi(\d+)ELi(\d+)E", name).group(2))
    total_stores = sum(1 for op, _ in ins if op == "global_store_dwordx4")
    if total_stores != maxm * stores_per_seg:
        return f"{name}: {total_stores} coefficient stores in the kernel, expected {maxm} x {stores_per_seg}"
    tail = ins[loads[-1] + 1:]
    n_store = sum(1 for op, _ in tail if op == "global_store_dwordx4")
    want = 2 * stores_per_seg
    contiguous = n_store == want          # otherwise the block layout interleaves earlier segments: skip check 3
    waits = {int(m) for op, a in ins if op == "s_waitcnt" for m in re.findall(r"vmcnt\((\d+)\)", a)}
    if want not in waits or stores_per_seg not in waits:
        return f"{name}: no s_waitcnt vmcnt({want}) / vmcnt({stores_per_seg}) in the kernel (found {sorted(waits)})"
    for op, a in (tail if contiguous else []):
        if op.startswith("global_load_dwordx") or op in ("s_endpgm",):
            continue
        ops = a.split(",")
        srcs = ",".join(ops[1:]) if op.startswith(("v_", "ds_read", "global_load")) else a
        if op.startswith(("global_store", "ds_write", "v_cmp", "v_cmpx", "s_")):
            srcs = a
        hit = regs(srcs) & dest
        if hit:
            return f"{name}: `{op} {a}` reads prefetch register(s) v{sorted(hit)} before the wait"
    return None if contiguous else ""


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "drone_path_planning_python_amd", "csrc",
                                                              "msnap_solve.o")
    text = disassemble(obj)
    kernels, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(_ZN5msnap16solve_kernel_reg\w+)>:", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif re.match(r"^[0-9a-f]+ <", ln):
            if not re.match(r"^[0-9a-f]+ <L\d+>:", ln):
                cur = None
        elif cur is not None:
            kernels[cur].append(ln.split("//")[0].strip())
    if len(kernels) != 4:
        print(f"check_prefetch_isa: expected 4 solve_kernel_reg instances, found {len(kernels)}")
        return 1
    res = {n: check_kernel(n, b) for n, b in sorted(kernels.items())}
    bad = [e for e in res.values() if e]
    for e in bad:
        print("check_prefetch_isa:", e)
    if not bad:
        full = sum(1 for e in res.values() if e is None)
        print(f"check_prefetch_isa: {len(kernels)} solve_kernel_reg instances ok (store counts and waits; "
              f"{full} of them also laid out contiguously: stores after the prefetch counted, no early reads)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
