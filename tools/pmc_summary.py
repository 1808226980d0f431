#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters: python tools/pmc_summary.py <counter_collection.csv> [name filter]"""
import csv
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
rows = defaultdict(lambda: defaultdict(list))
meta = {}
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:100]
        if flt and flt not in k:
            continue
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows[k]["~duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
for k, c in sorted(rows.items()):
    print(f"{k}  vgpr/agpr/sgpr/lds/grid/wg = {meta[k]}")
    for name, v in sorted(c.items()):
        print(f"    {name:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")
