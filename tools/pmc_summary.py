#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch over decode-sized grids).
Usage: python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections, csv, glob, sys
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fh::", "")
        if filt and filt not in name: continue
        key = f"{name} grid={r['Grid_Size']} wg={r['Workgroup_Size']}"
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    n = max(len(v) for v in cs.values())
    print(f"{k}  dispatches={n}")
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} mean={sum(v)/len(v):16.1f}")
