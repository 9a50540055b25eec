#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7) rocpd sqlite database: per-kernel launches, avg/min/max
duration, registers and LDS -- the same content as `--stats` kernel_stats, plus per-grid-size
rows for kernels launched once per pyramid level.  Usage: rocpd_stats.py results.db [--by-grid]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = "--by-grid" in sys.argv
    cur = db.cursor()
    key = "s.kernel_name" + (", d.grid_size_x, d.grid_size_y, d.grid_size_z" if by_grid else "")
    q = ("select %s, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), sum(d.end - d.start), "
         "max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by %s "
         "order by sum(d.end - d.start) desc" % (key, key))
    rows = cur.execute(q).fetchall()
    tot = sum(r[-4] for r in rows) or 1
    print("%-58s %8s %10s %10s %10s %7s %5s %5s %7s" % ("kernel" + (" [grid]" if by_grid else ""), "calls", "avg_us", "min_us", "max_us", "share", "vgpr", "sgpr", "lds_B"))
    for r in rows:
        name = r[0].split("(")[0][-50:]
        if by_grid:
            name += " [%dx%dx%d]" % (r[1], r[2], r[3])
            r = (r[0],) + r[4:]
        print("%-58s %8d %10.2f %10.2f %10.2f %6.1f%% %5d %5d %7d" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot, r[6], r[7], r[8]))
    span = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print("total kernel time %.3f ms over a %.3f ms span" % (tot / 1e6, (span[1] - span[0]) / 1e6))


if __name__ == "__main__":
    main()
