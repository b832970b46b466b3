#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel and grid size, the
count / median / mean / min duration.  Usage: prof_summary.py <dir-or-csv> [out.md]"""
import csv
import glob
import os
import statistics
import sys


def main():
    src = sys.argv[1]
    paths = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    groups = {}
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                short = name.split("(")[0].replace("void ", "")
                if len(short) > 70:
                    short = short[:67] + "..."
                key = (short, int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1), int(r["Workgroup_Size_X"]),
                       int(r["LDS_Block_Size"]), int(r["VGPR_Count"]), int(r["SGPR_Count"]))
                groups.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines = ["| kernel | grid (threads) | wg | LDS B | VGPR | SGPR | calls | median us | mean us | min us | max us |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for key, d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| `%s` | %d | %d | %d | %d | %d | %d | %.2f | %.2f | %.2f | %.2f |" % (
            key[0], key[1], key[2], key[3], key[4], key[5], len(d), statistics.median(d) / 1e3,
            statistics.mean(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
