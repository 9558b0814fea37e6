#!/usr/bin/env python3
"""Per-kernel summary of the largest forward in a rocprofv3 --kernel-trace sqlite/csv output (development aid).
Usage: trace_summary.py <results.db> [min_launches] [segment_index]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name,d.start,d.end,d.grid_size_x,d.grid_size_y,d.grid_size_z,d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
segs, cur = [], [rows[0]]
for r in rows[1:]:
    if r[1] - cur[-1][2] > 300000: segs.append(cur); cur = []
    cur.append(r)
segs.append(cur)
minl = int(sys.argv[2]) if len(sys.argv) > 2 else 300
big = [s for s in segs if len(s) > minl]
print('segments (launches, wall us):', [(len(x), round((x[-1][2] - x[0][1]) / 1e3)) for x in big])
s = big[int(sys.argv[3])] if len(sys.argv) > 3 else big[min(2, len(big) - 1)]
agg = {}
for r in s:
    n = re.sub(r'\(.*', '', r[0]).replace('void ', '')
    a = agg.setdefault((n[:72], r[3], r[4], r[5], r[6]), [0, 0]); a[0] += 1; a[1] += r[2] - r[1]
tot = sum(a[1] for a in agg.values())
print(f"launches {len(s)} wall {(s[-1][2]-s[0][1])/1e3:.1f} us, sum {tot/1e3:.1f} us")
for k, a in sorted(agg.items(), key=lambda x: -x[1][1])[:18]:
    print(f"{k[0]:72s} {k[1]}x{k[2]}x{k[3]} wg{k[4]} n={a[0]} avg={a[1]/a[0]/1e3:.1f} tot={a[1]/1e3:.0f} {100*a[1]/tot:.1f}%")
