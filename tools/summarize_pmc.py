#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs (one directory per pass) -> one JSON + Markdown table.

    python tools/summarize_pmc.py OUT.json OUT.md label=dir [label=dir ...]

Derived columns (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_16x16x4_f32; GRBM_GUI_ACTIVE
is summed over the 8 XCDs; FETCH_SIZE is in KiB and reads half the bytes of wide coalesced reads on gfx950 -> doubled):
    mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
    v          = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA     non-MFMA VALU instructions per MFMA
    hbm_bytes  = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(dict)
        with open(path) as fh:
            for row in csv.DictReader(fh):
                key = (row["Kernel_Name"], row["Dispatch_Id"])
                per_dispatch[key][row["Counter_Name"]] = per_dispatch[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                per_dispatch[key]["_grid"] = float(row["Grid_Size"])
                per_dispatch[key]["_ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        for (kname, _), counters in per_dispatch.items():
            for c, v in counters.items():
                acc[short(kname)][c].append(v)
    return acc


def main():
    out_json, out_md, specs = sys.argv[1], sys.argv[2], sys.argv[3:]
    result = {}
    for spec in specs:
        label, d = spec.split("=", 1)
        acc = load(d)
        for k, counters in acc.items():
            if not k.startswith("k_"):
                continue
            # a kernel may be launched with several grids (workloads): keep them apart
            grids = sorted(set(counters["_grid"]))
            for gsz in grids:
                idx = [i for i, g_ in enumerate(counters["_grid"]) if g_ == gsz]
                if len(idx) < 1:
                    continue
                row = result.setdefault(f"{k} grid={int(gsz)}", {"dispatches": {}, "label": label})
                for c, vals in counters.items():
                    sel = [vals[i] for i in idx if i < len(vals)]
                    if sel and not c.startswith("_"):
                        row[c] = sum(sel) / len(sel)
                row["dispatches"][label] = len(idx)
                row.setdefault("ns", {})[label] = sum(counters["_ns"][i] for i in idx) / len(idx)
    for k, r in result.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in r and r.get("GRBM_GUI_ACTIVE"):
            r["mfma_util"] = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * r["GRBM_GUI_ACTIVE"] / 8.0)
        if r.get("SQ_INSTS_MFMA"):
            r["v"] = (r.get("SQ_INSTS_VALU", 0.0) - r["SQ_INSTS_MFMA"]) / r["SQ_INSTS_MFMA"]
        if "FETCH_SIZE" in r or "WRITE_SIZE" in r:
            r["hbm_bytes"] = 1024.0 * (2.0 * r.get("FETCH_SIZE", 0.0) + r.get("WRITE_SIZE", 0.0))
    json.dump(result, open(out_json, "w"), indent=1, sort_keys=True)
    cols = ["mfma_util", "v", "hbm_bytes", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_ANY",
            "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM_RD"]
    with open(out_md, "w") as fh:
        fh.write("| kernel | " + " | ".join(cols) + " | kernel us (profiled) |\n|---|" + "---|" * (len(cols) + 1) + "\n")
        for k in sorted(result):
            r = result[k]
            ns = list(r.get("ns", {}).values())
            fh.write(f"| `{k}` | " + " | ".join(f"{r[c]:.4g}" if c in r else "" for c in cols) + f" | {sum(ns) / len(ns) / 1e3:.1f} |\n")
    print(open(out_md).read())


if __name__ == "__main__":
    main()
