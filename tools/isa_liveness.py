#!/usr/bin/env python3
"""Approximate VGPR liveness over one kernel of a gfx950 assembly listing: where the register pressure peaks and
which definitions are still live there.  (This is how the deferred input / pivot checks of solve_kernel_twin were
found: 50 registers of |w|, |T| and pivots waiting for compares the scheduler had sunk behind both sweeps.)

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -S --cuda-device-only -o solve.s csrc/msnap_solve.hip
    python3 tools/isa_liveness.py solve.s _ZN5msnap17solve_kernel_twinILi5ELi10EEEvPKdS2_iiPdS3_Pii [lo hi]

The first operand of v_* (except v_cmp*), ds_read*, global_load*, scratch_load* counts as a definition, every other
VGPR operand as a use; a physical register's live range runs from a definition to its last use before the next one
(accumulating forms -- v_fmac, v_cndmask, DPP -- also read their destination).  With [lo hi] it lists the
definitions inside that instruction range that are still live at the peak, with their last use."""
import collections
import re
import sys


def kernel_body(path, symbol):
    out, on = [], False
    for line in open(path).read().splitlines():
        if line.startswith(symbol + ":"):
            on = True
            continue
        if on:
            t = line.split(";")[0].strip()
            if t.startswith("s_endpgm"):
                break
            if not t or t.endswith(":") or t.startswith("."):
                continue
            out.append(t)
    return out


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def main():
    ins = kernel_body(sys.argv[1], sys.argv[2])
    if not ins:
        sys.exit("kernel symbol not found")
    last_def, last_use, intervals = {}, {}, []
    for k, t in enumerate(ins):
        p = t.split(None, 1)
        op, args = p[0], (p[1] if len(p) > 1 else "")
        ops = [x.strip() for x in args.split(",")]
        isdef = op.startswith(("v_", "ds_read", "global_load", "scratch_load", "buffer_load")) and not op.startswith(("v_cmp", "v_cmpx"))
        d = regs(ops[0]) if (isdef and ops) else set()
        u = regs(",".join(ops[1:])) if isdef else regs(args)
        if op.startswith(("v_fmac", "v_mac", "v_cndmask")) or "dpp" in op:
            u |= d
        for r in u:
            last_use[r] = k
        for r in d:
            if r in last_def and r in last_use and last_use[r] >= last_def[r]:
                intervals.append((last_def[r], last_use[r], r))
            last_def[r] = k
            last_use.pop(r, None)
    for r in last_def:
        if r in last_use and last_use[r] >= last_def[r]:
            intervals.append((last_def[r], last_use[r], r))
    n = len(ins)
    delta = [0] * (n + 2)
    for a, b, _ in intervals:
        delta[a] += 1
        delta[b + 1] -= 1
    live, c = [], 0
    for k in range(n):
        c += delta[k]
        live.append(c)
    peak = max(live)
    at = live.index(peak)
    print(f"{n} instructions, peak {peak} live VGPRs at instruction {at}: {ins[at][:70]}")
    step = max(1, n // 50)
    for k in range(0, n, step):
        print(f"{k:6d} {live[k]:4d}  {ins[k][:60]}")
    held = sorted((a, b, r) for a, b, r in intervals if a <= at <= b)
    print("definitions live at the peak, by position:", sorted(collections.Counter(a // step * step for a, _, _ in held).items()))
    if len(sys.argv) > 4:
        lo, hi = int(sys.argv[3]), int(sys.argv[4])
        seen = set()
        for a, b, _ in held:
            if lo <= a < hi and (a, b) not in seen:
                seen.add((a, b))
                print(f"{a:6d} -> {b:6d}  {ins[a][:64]:64s} | last use: {ins[b][:56]}")


if __name__ == "__main__":
    main()
