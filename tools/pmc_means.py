#!/usr/bin/env python3
"""Reduce a rocprofv3 `*_counter_collection.csv` (one row per dispatch and counter) to one row per (kernel, counter):

    python3 tools/pmc_means.py <counter_collection.csv>  >  profiles/<name>_pmc_<COUNTERS>.csv

Output columns: Kernel_Name, Counter_Name, Mean, Min, Max, Dispatches, Grid_Size, Workgroup_Size, VGPR_Count,
LDS_Block_Size.  FETCH_SIZE / WRITE_SIZE are in KB per dispatch as rocprofv3 reports them (FETCH_SIZE counts 32-byte
units as 64 on gfx950's TCC: multiply by 2, /opt/skills/guides/MI355X_MICROARCH.md)."""
import csv
import sys
from collections import OrderedDict


def reduce(path):
    acc = OrderedDict()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = (r["Kernel_Name"], r["Counter_Name"])
            v = float(r["Counter_Value"])
            e = acc.setdefault(k, dict(n=0, s=0.0, lo=v, hi=v, grid=r.get("Grid_Size", ""), wg=r.get("Workgroup_Size", ""),
                                       vgpr=r.get("VGPR_Count", ""), lds=r.get("LDS_Block_Size", "")))
            e["n"] += 1; e["s"] += v; e["lo"] = min(e["lo"], v); e["hi"] = max(e["hi"], v)
    return acc


def main():
    acc = reduce(sys.argv[1])
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Kernel_Name", "Counter_Name", "Mean", "Min", "Max", "Dispatches", "Grid_Size", "Workgroup_Size",
                "VGPR_Count", "LDS_Block_Size"])
    for (kern, ctr), e in acc.items():
        w.writerow([kern, ctr, round(e["s"] / e["n"], 3), e["lo"], e["hi"], e["n"], e["grid"], e["wg"], e["vgpr"], e["lds"]])


if __name__ == "__main__":
    main()
