#!/usr/bin/env python3
"""Build-time tripwire: no register-pressure copy (v_accvgpr_write_b32, scratch store) under a reduced exec mask.

Why: the compiler keeps values that do not fit the 256 architectural VGPRs in accumulation registers
(or scratch) and inserts the copies where its live-range splitting puts them -- block boundaries of the control-flow
graph, with no regard for exec.  Round 4 found `v_accvgpr_write_b32 a2, v22` / `a4, v18` (lane + 64, lane + 128 of the
cross-tile prefetch) in the EXIT block of a loop the lanes leave one by one, in front of the `s_or_b64 exec, exec, ...`
that restores the mask: exec == 0 there, the copies wrote no lane, and the prefetch of solve_kernel_twin<5, 20> read
at base + 16 * garbage -- `Memory access fault by GPU`.  Round 3's abort of the two-sided 16-segment order-9 instance
(188 bytes of scratch) has the same shape with scratch stores.

The check walks every kernel's control-flow graph with an abstract exec state: `s_*_saveexec_b64 pair` pushes the
pair that holds the wider mask, `s_or_b64 exec, exec, pair` pops down to it, `s_andn2_b64 exec, exec, ..` (lanes
leaving a loop) marks the mask narrow until the next restore.  A copy reached with a non-empty stack or a narrow mask
is reported.  Conservative: a copy of a value that lives only inside a divergent region would be reported too.
Second rule, for kernels without such copies (an ordinary v_mov placed the same way looks like any other move): behind
the latch of a loop that runs until no lane is left (`s_cbranch_execnz` backwards) exec is zero on EVERY execution until
it is written again, so any vector instruction there is reported -- it can only be one the compiler misplaced.

    python tools/check_exec_isa.py [object files ...]        (default: csrc/msnap_solve.o msnap_aux.o msnap_grid.o)
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "drone_path_planning_python_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"

SPILL = re.compile(r"^(v_accvgpr_write_b32|scratch_store_\w+|buffer_store_\w+ .*\boffen\b)")
PAIR = r"(s\[\d+:\d+\]|vcc)"
PUSH = re.compile(rf"^s_(?:and|andn2|or|xor|nand|nor)_saveexec_b64 {PAIR},")
OR_EXEC = re.compile(rf"^s_or_b64 exec, exec, {PAIR}")
ALIAS = re.compile(rf"^s_(?:xor_b64 {PAIR}, exec, {PAIR}|mov_b64 {PAIR}, {PAIR})$")    # the saved mask changes its register
MOV_EXEC = re.compile(rf"^s_mov_b64 exec, (\S+)")
NARROW = re.compile(r"^s_(?:andn2|and)_b64 exec, exec,")       # lanes leave (loops with a divergent trip count)
EXEC_WRITE = re.compile(r"^s_\w+_saveexec_b64|^s_\w+ exec\b|^v_cmpx")
VECTOR = re.compile(r"^(v_(?!readlane|readfirstlane|writelane)|ds_|global_|flat_|buffer_|scratch_)")
OTHER_EXEC = re.compile(r"^s_\w+ exec\b|^v_cmpx")               # any other writer of exec: treated as narrowing


def disassemble(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat"), os.path.join(tmp, "co")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(tmp, "copy.o")], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}"], check=True)
        text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
    for filt in ("c++filt", f"{LLVM}/llvm-cxxfilt"):      # readable kernel names; the check itself does not need them
        try:
            return subprocess.run([filt], input=text, capture_output=True, text=True, check=True).stdout
        except (OSError, subprocess.CalledProcessError):
            continue
    return text


def kernels_of(text):
    """name -> list of (instruction text, address, branch target address or None)"""
    out, cur, base = {}, None, 0
    for ln in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:\s*$", ln)
        if m:
            cur, base = m.group(2), int(m.group(1), 16)
            out[cur] = []
            continue
        if cur is None or "//" not in ln:
            continue
        ins = ln.split("//")[0].strip()
        a = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
        if not ins or not a:
            continue
        t = re.search(r"\+0x([0-9a-fA-F]+)>\s*$", ln)      # (the demangled name itself holds '<' and '>')
        out[cur].append((ins, int(a.group(1), 16), base + int(t.group(1), 16) if t else None))
    return out


def walk(body):
    """Abstract interpretation of exec over the kernel's control-flow graph.  State: the stack of SGPR pairs that hold
    a saved (wider) mask, plus `narrow` (lanes have left through s_andn2 exec since the last restore).  Structured
    control flow as LLVM emits it: s_*_saveexec pushes, `s_or_b64 exec, exec, pair` pops down to that pair.
    Returns the list of (instruction, state) for register-pressure copies executed with a possibly reduced mask."""
    at = {addr: i for i, (_, addr, _) in enumerate(body)}
    seen, bad, work = set(), [], [(0, (), False, False)]
    while work:
        i, stack, narrow, zero = work.pop()
        while i < len(body):
            key = (i, stack, narrow, zero)
            if key in seen:
                break
            seen.add(key)
            ins, _, target = body[i]
            if SPILL.match(ins) and (stack or narrow):
                bad.append((ins, stack, narrow))
            elif zero and VECTOR.match(ins):
                # exec is zero here on every execution (behind the latch of a loop that runs until no lane is left, no
                # write of exec since): a vector instruction does nothing -- if the compiler put one here it believes
                # otherwise (the copies of round 4 sat exactly here)
                bad.append((ins, ("<exec == 0>",), False))
            if EXEC_WRITE.match(ins):
                zero = False
            m = PUSH.match(ins)
            if m:       # (the same pair again: the else flip of an if, or the region re-entered in a loop)
                pair = m.group(1)
                stack = (stack[:stack.index(pair)] if pair in stack else stack) + (pair,)
                if len(stack) > 8:
                    if any(SPILL.match(b[0]) for b in body):
                        bad.append((ins, ("<nesting not understood>",), False))
                    break
            elif ALIAS.match(ins):
                g = ALIAS.match(ins).groups()
                dst, src = (g[0], g[1]) if g[0] else (g[2], g[3])
                if src in stack and dst != "exec":      # `s_xor sA, exec, sB`: sA = the lanes to come back (else part)
                    stack = tuple(dst if p == src else p for p in stack)
            elif OR_EXEC.match(ins):
                pair = OR_EXEC.match(ins).group(1)
                if pair in stack:
                    stack = stack[:stack.index(pair)]
                    narrow = False
                else:               # a loop's accumulated mask of the lanes that left it: back to the loop's entry mask
                    narrow = False
            elif MOV_EXEC.match(ins):
                src = MOV_EXEC.match(ins).group(1)
                if src in stack:
                    stack, narrow = stack[:stack.index(src)], False
                elif src != "-1":
                    narrow = True
                else:
                    stack, narrow = (), False
            elif NARROW.match(ins) or OTHER_EXEC.match(ins):
                narrow = True
            op = ins.split()[0]
            if op == "s_endpgm":
                break
            if op == "s_branch":
                if target is None or target not in at:
                    break
                i = at[target]
                continue
            if op.startswith("s_cbranch") and target is not None and target in at:
                work.append((at[target], stack, narrow, zero))
                if op == "s_cbranch_execnz" and at[target] <= i:      # the latch of a loop the lanes leave one by one:
                    zero = True                                        # behind it exec is zero on EVERY execution
            i += 1
    visited = {i for i, _, _, _ in seen}
    missed = [body[i][0] for i in range(len(body)) if i not in visited and SPILL.match(body[i][0])]
    if missed:      # a copy the walk never reached: the graph was not understood -- report rather than pass
        bad.append((missed[0], ("<unreached>",), False))
    return bad


def check(obj):
    bad, n_spill = [], 0
    ks = kernels_of(disassemble(obj))
    for name, body in ks.items():
        n_spill += sum(1 for ins, _, _ in body if SPILL.match(ins))
        seen_ins = set()
        for ins, stack, narrow in walk(body):
            if ins not in seen_ins:
                seen_ins.add(ins)
                bad.append((name.split("(")[0], ins, "lanes left a loop" if narrow and not stack else "inside " + " > ".join(stack)))
    return len(ks), n_spill, bad


def main():
    objs = sys.argv[1:] or [os.path.join(CSRC, f) for f in ("msnap_solve.o", "msnap_aux.o", "msnap_grid.o")]
    rc = 0
    for obj in objs:
        n_kernels, n_spill, bad = check(obj)
        for k, ins, since in bad:
            print(f"check_exec_isa: {os.path.basename(obj)}: {k}: `{ins}` under a reduced exec mask ({since})")
        if bad:
            rc = 1
        else:
            print(f"check_exec_isa: {os.path.basename(obj)}: {n_kernels} kernels, {n_spill} register-pressure copies, "
                  "none under a reduced exec mask")
    return rc


if __name__ == "__main__":
    sys.exit(main())
