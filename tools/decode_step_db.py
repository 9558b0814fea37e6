#!/usr/bin/env python3
"""One steady-state decode step per decode run, cut from a rocprofv3 --kernel-trace results.db (rocpd sqlite):
a step = the launches between two argmax_final_kernel launches (the last launch of a decode step: it also advances the
device-side decode state); consecutive steps with the same launch count form a run
(bench.py: the c=32 window, then the c=1/4/16 sweep).  Prints, per run, launches per step, wall time of a late step
and the per-kernel mean durations inside it.  Usage: decode_step_db.py <results.db> [min_steps_in_run]"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, accum_vgpr_count from kernels order by start"))
    adv = [i for i, r in enumerate(rows) if "argmax_final" in r[0]]
    min_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    runs, cur = [], [adv[0]]
    for a, b in zip(adv, adv[1:]):
        if rows[b][1] - rows[a][2] > 3_000_000 * 4 or (b - a) != (cur[-1] - cur[-2] if len(cur) > 1 else b - a):
            runs.append(cur)
            cur = []
        cur.append(b)
    runs.append(cur)
    for run in runs:
        if len(run) < min_steps:
            continue
        lo, hi = run[-3], run[-2]
        seg = rows[lo + 1:hi + 1]
        wall = (seg[-1][2] - seg[0][1]) / 1e3
        steps_wall = [(rows[b][2] - rows[a][2]) / 1e3 for a, b in zip(run, run[1:])]
        by = collections.OrderedDict()
        busy = 0.0
        for r in seg:
            d = (r[2] - r[1]) / 1e3
            busy += d
            name = re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "")).replace("void ", "").replace("fh::", "")
            by.setdefault(f"{name[:78]} g={r[3]}x{r[4]}x{r[5]} wg={r[6]} vgpr={r[7]}+{r[8]}", []).append(d)
        med = sorted(steps_wall)[len(steps_wall) // 2]
        print(f"\n== decode run of {len(run)} steps: {len(seg)} launches per step, late-step wall {wall:.1f} us (median step {med:.1f} us), "
              f"sum of kernel durations {busy:.1f} us, gaps {wall - busy:.1f} us")
        print(f"{'kernel':118s} {'n':>4s} {'avg_us':>8s} {'total_us':>9s} {'%':>5s}")
        for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            print(f"{k[:118]:118s} {len(v):4d} {sum(v) / len(v):8.2f} {sum(v):9.1f} {100 * sum(v) / wall:5.1f}")


if __name__ == "__main__":
    main()
