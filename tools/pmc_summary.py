#!/usr/bin/env python3
"""Summarise the --pmc passes of tools/make_profiles.sh: per kernel and grid size the
mean of every counter, plus the HBM traffic per launch corrected as
MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE reads exactly half of a wide coalesced streaming read, so it is
doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Writes <dir>/pmc_traffic.json and prints a markdown table."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "msnap" not in name:
                    continue
                acc[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    traffic = {}
    print("| kernel | grid threads | counter | launches | mean per launch |")
    print("|---|---|---|---|---|")
    for (name, grid), ctrs in sorted(acc.items()):
        for c, v in sorted(ctrs.items()):
            print(f"| `{name}` | {grid} | {c} | {len(v)} | {sum(v) / len(v):.6g} |")
        if "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
            fetch = sum(ctrs["FETCH_SIZE"]) / len(ctrs["FETCH_SIZE"]) * 1024.0
            write = sum(ctrs["WRITE_SIZE"]) / len(ctrs["WRITE_SIZE"]) * 1024.0
            drones = grid // 4
            traffic[f"{name}|{grid}|M{os.environ.get('MSNAP_PROFILE_SEGMENTS', '10')}"] = {
                "drones": drones, "fetch_bytes_raw": fetch, "fetch_bytes_corrected_x2": 2 * fetch,
                "write_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write,
            }
    print()
    print("HBM traffic per launch (FETCH_SIZE x2 correction applied):")
    for k, v in traffic.items():
        print(f"- `{k}`: read {v['fetch_bytes_corrected_x2'] / 1e6:.3f} MB (raw counter {v['fetch_bytes_raw'] / 1e6:.3f} MB), "
              f"write {v['write_bytes'] / 1e6:.3f} MB, total {v['hbm_bytes_per_launch'] / 1e6:.3f} MB "
              f"for {v['drones']} drones = {v['hbm_bytes_per_launch'] / max(v['drones'], 1):.0f} B/trajectory")
    with open(os.path.join(root, "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)


if __name__ == "__main__":
    main()
