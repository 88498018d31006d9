#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite) kernel trace the way `--stats` does:
per kernel name: calls, total, average, min, max duration (ns) and share.
usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"']
    for n, k, s, a, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % (n.replace('"', "'"), k, s, a, mn, mx, 100.0 * s / tot))
    out = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out)
    sys.stdout.write(out)


if __name__ == '__main__':
    main()
