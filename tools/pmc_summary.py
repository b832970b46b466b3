#!/usr/bin/env python3
"""Summarise the --pmc passes of tools/make_profiles.sh.

Directories are named pmc_<workload>_<what>; every msnap kernel gets a table row per
(kernel, grid, counter) with the mean over launches.  For the workload-tagged
FETCH_SIZE / WRITE_SIZE passes (<workload> = "<drones>x<seg>o<order>") the HBM traffic per
launch of the dominant solve kernel is written to <dir>/pmc_traffic.json, corrected as
MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KB; on gfx950
FETCH_SIZE reads exactly half of a wide coalesced streaming read, so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_path_planning_python_amd._lib import csrc_sha  # noqa: E402


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))        # (kernel, grid) -> counter -> values
    per_wl = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))   # workload -> kernel -> counter -> values
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        wl = os.path.basename(d)[4:].split("_")[0]
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    if "msnap" not in name:
                        continue
                    v = float(r["Counter_Value"])
                    acc[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(v)
                    per_wl[wl][name][r["Counter_Name"]].append(v)
    print("| kernel | grid threads | counter | launches | mean per launch |")
    print("|---|---|---|---|---|")
    for (name, grid), ctrs in sorted(acc.items()):
        for c, v in sorted(ctrs.items()):
            print(f"| `{name}` | {grid} | {c} | {len(v)} | {sum(v) / len(v):.6g} |")
    traffic = {}
    for wl, kernels in per_wl.items():
        if "x" not in wl or "o" not in wl:
            continue
        # "<drones>x..": the per-drone solve (K1); "grid<drones>x..": the shared-grid GEMM (K2) of the same batch
        want = "grid_gemm" if wl.startswith("grid") else "solve_kernel"
        cands = {k: c for k, c in kernels.items() if want in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c}
        if not cands:
            continue
        name = max(cands, key=lambda k: len(cands[k]["FETCH_SIZE"]))
        c = cands[name]
        fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024.0
        write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024.0
        drones = int(wl.split("x")[0].replace("grid", ""))
        traffic[wl] = {"kernel": name, "drones": drones, "csrc_sha": csrc_sha(), "fetch_bytes_raw": fetch,
                       "fetch_bytes_corrected_x2": 2 * fetch, "write_bytes": write,
                       "hbm_bytes_per_launch": 2 * fetch + write}
    print()
    print("HBM traffic per launch (FETCH_SIZE x2 correction applied):")
    for k, v in traffic.items():
        print(f"- `{k}` ({v['kernel']}): read {v['fetch_bytes_corrected_x2'] / 1e6:.3f} MB (raw counter "
              f"{v['fetch_bytes_raw'] / 1e6:.3f} MB), write {v['write_bytes'] / 1e6:.3f} MB, total "
              f"{v['hbm_bytes_per_launch'] / 1e6:.3f} MB for {v['drones']} drones = "
              f"{v['hbm_bytes_per_launch'] / max(v['drones'], 1):.0f} B/trajectory")
    if traffic:
        with open(os.path.join(root, "pmc_traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)


if __name__ == "__main__":
    main()
