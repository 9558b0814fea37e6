#!/usr/bin/env python3
"""Median duration per (kernel, grid, workgroup) of a rocprofv3 --kernel-trace results.db, in first-dispatch order
(development aid for form sweeps).  Usage: kernel_durations_db.py <results.db> [name-substring]"""
import re, sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name,d.start,d.end,d.grid_size_x,d.grid_size_y,d.grid_size_z,d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start")
agg, order = {}, []
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for name, t0, t1, gx, gy, gz, wg in rows:
    n = re.sub(r'\(.*', '', name).replace('void ', '').replace('fh::', '')
    if sub and sub not in n: continue
    key = (n[:60], gx // wg, gy, gz, wg)
    if key not in agg: agg[key] = []; order.append(key)
    agg[key].append((t1 - t0) / 1e3)
for k in order:
    v = agg[k]
    tail = v[len(v) // 3:]
    print(f"{k[0]:60s} wgs={k[1]:5d}x{k[2]}x{k[3]:<3d} wg={k[4]:4d} n={len(v):3d} med={statistics.median(tail):7.2f} min={min(v):7.2f}")
