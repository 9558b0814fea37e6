#!/usr/bin/env python3
"""Summarise ONE steady-state decode step from a rocprofv3 --kernel-trace CSV: per-kernel launches, mean and
total device time inside the step (between two decode_advance_kernel launches).  Usage:
    python tools/decode_step_profile.py <kernel_trace.csv> [step_index_from_end]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    adv = [i for i, r in enumerate(rows) if "decode_advance" in r["Kernel_Name"]]
    lo, hi = adv[-back - 1], adv[-back]
    seg = rows[lo + 1:hi + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    by = collections.defaultdict(list)
    busy = 0.0
    for r in seg:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        busy += d
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fh::", "")
        by[f"{name} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg={r['Workgroup_Size_X']} vgpr={r['VGPR_Count']}"].append(d)
    print(f"decode step: {len(seg)} kernel launches, wall {(t1 - t0) / 1e3:.1f} us, sum of kernel durations {busy:.1f} us")
    print(f"{'kernel':110s} {'n':>4s} {'avg_us':>8s} {'total_us':>9s} {'%':>5s}")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[:110]:110s} {len(v):4d} {sum(v) / len(v):8.2f} {sum(v):9.1f} {100 * sum(v) / busy:5.1f}")


if __name__ == "__main__":
    main()
