#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size) statistics, so that several workloads of ONE kernel instantiation (tools/pmc_targets.py
tags them by the number of workgroups) are not averaged together as `--stats` does.  Also lists VGPR / LDS per kernel.

    python tools/kernel_trace_by_grid.py <*_kernel_trace.csv> <out.csv>"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:70]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(list)
    meta = {}
    with open(src) as fh:
        for row in csv.DictReader(fh):
            key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]), int(row["Workgroup_Size_X"]))
            acc[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta[key] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"])
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "Grid", "Workgroup", "Calls", "AverageNs", "MinNs", "MaxNs", "VGPR", "AccumVGPR", "SGPR", "LDS_Block_Size"])
        for key in sorted(acc, key=lambda k: -sum(acc[k])):
            d = acc[key]
            w.writerow([key[0], key[1], key[2], len(d), round(sum(d) / len(d), 1), min(d), max(d), *meta[key]])
    print(open(dst).read())


if __name__ == "__main__":
    main()
