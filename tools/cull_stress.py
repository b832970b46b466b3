#!/usr/bin/env python3
"""Randomised check of the pairwise pass's broad phase against the C oracle: random swarm sizes (also one short of and one
past the tile sizes), sample counts, swarm kinds (dense, sparse, lattice teams with ties, beyond the sort lattice), NaN
drones and samples, every evaluator mode.  usage: cull_stress.py [seconds] [seed].  GPU box only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import c_oracle  # noqa: E402
from drone_path_planning_python_amd import Context  # noqa: E402
from test_formation_full import _broad_phase_swarm  # noqa: E402


def run(budget=60.0, seed=1, max_cases=None, big=True, others=True):
    rng = np.random.default_rng(seed)
    edges = [256, 257, 511, 512, 513, 1023, 1025, 2047, 2049, 3071, 3073, 4095, 4097, 6145, 8191, 8193]
    t0, n_cases = time.time(), 0
    with Context(order=7, max_segments=16) as ctx:
        ctx.set_option("collide_cull_min_drones", 256)
        while time.time() - t0 < budget and (max_cases is None or n_cases < max_cases):
            n = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(256, 7000))
            S = int(rng.integers(6, 40))
            if big and rng.random() < 0.03:      # the largest swarms the broad phase takes (seconds of oracle each)
                n, S = int(rng.choice([12289, 16383, 16384])), int(rng.integers(6, 10))
            kind = str(rng.choice(["dense", "sparse", "teams", "far"]))
            pos = _broad_phase_swarm(kind, n, S, rng)
            for _ in range(int(rng.integers(0, 4))):
                pos[int(rng.integers(0, n))] = np.nan
            for _ in range(int(rng.integers(0, 4))):
                pos[int(rng.integers(0, n)), int(rng.integers(0, S))] = np.nan
            if rng.random() < 0.5:
                pos[int(rng.integers(0, n))] = pos[int(rng.integers(0, n))]
            mode = int(rng.integers(0, 3))
            parts = int(rng.choice([0, 0, 1, 2, 3, 5, 8]))
            ctx.set_option("collide_cull_mode", mode)
            ctx.set_option("collide_sample_parts", parts)
            ref = c_oracle.formation_collide(pos, 0.3)
            # the other ways through the pass, now and then: broad phase off, a shard of the rows, the pass in parts
            way = str(rng.choice(["whole", "whole", "whole", "plain", "shard", "parts"])) if others else "whole"
            if way == "plain":
                ctx.set_option("collide_no_cull", 1)
                got = ctx.formation_collide(pos, pos, 0.3)
                ctx.set_option("collide_no_cull", 0)
                assert ctx.get_option("collide_last_cull") == 0
            elif way == "shard":
                lo = int(rng.integers(0, n - 1))
                cnt = int(rng.integers(1, n - lo + 1))
                got = ctx.formation_collide(pos[lo:lo + cnt], pos, 0.3, row_offset=lo)
                ref = tuple(x[lo:lo + cnt] for x in ref)
            elif way == "parts":
                P = int(rng.integers(2, 10))
                ctx.set_option("collide_sample_parts", int(rng.choice([0, 0, 4])))
                blocks = np.stack([ctx.formation_collide_part(pos, q, P) for q in range(P)])
                got = ctx.formation_collide_finish(blocks, n, 0.3)
            else:
                got = ctx.formation_collide(pos, pos, 0.3)
                assert ctx.get_option("collide_last_cull") == 1
            mode = f"{mode} way={way}"
            for a, b, what in zip(got, ref, ("min_dist", "partner", "hit")):
                if not np.array_equal(a, b):
                    bad = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0][:5]
                    print(f"MISMATCH {what}: n={n} S={S} kind={kind} mode={mode} parts={parts} rows {bad} got {a[bad]} want {b[bad]}")
                    print("nan drones", np.nonzero(np.isnan(pos).all(axis=(1, 2)))[0], "bad rows total", int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum()))
                    for m2 in ((0, 1, 2) if "whole" in str(mode) else ()):
                        for p2 in (0, 1, 2, 3, 5, 8):
                            ctx.set_option("collide_cull_mode", m2)
                            ctx.set_option("collide_sample_parts", p2)
                            g2 = ctx.formation_collide(pos, pos, 0.3)
                            print(f"  mode {m2} parts {p2}: equal = {all(np.array_equal(x, y) for x, y in zip(g2, ref))}")
                    ctx.set_option("collide_no_cull", 1)
                    g2 = ctx.formation_collide(pos, pos, 0.3)
                    print("  no cull: equal =", all(np.array_equal(x, y) for x, y in zip(g2, ref)))
                    np.save(os.path.join(ROOT, "gpurun_out", "cull_stress_fail.npy"), pos)
                    sys.exit(1)
            n_cases += 1
    print(f"cull_stress: {n_cases} random cases equal to the oracle in {time.time() - t0:.0f} s")
    return n_cases


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
