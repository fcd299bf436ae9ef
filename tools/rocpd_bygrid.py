#!/usr/bin/env python3
"""Group the dispatches of kernels whose name contains SUBSTR by launch grid (workgroups) in a rocprofv3 rocpd
database: which problem shapes the time of one kernel template goes to.
usage: python tools/rocpd_bygrid.py results.db SUBSTR [steps]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    sub = sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
    gx = "d.grid_size_x" if "grid_size_x" in cols else "d.grid_x"
    wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else "d.workgroup_x"
    gy = gx.replace("_x", "_y")
    gz = gx.replace("_x", "_z")
    rows = db.execute(
        "select s.kernel_name, %s/%s, %s, %s, count(*), sum(d.end-d.start) from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like ? "
        "group by 1,2,3,4 order by 6 desc" % (gx, wx, gy, gz), ("%" + sub + "%",)).fetchall()
    print("%-44s %8s %5s %5s %7s %10s %9s" % ("kernel", "blocks_x", "gy", "gz", "calls/st", "ms/step", "avg_us"))
    for name, bx, y, z, n, tot in rows:
        print("%-44s %8d %5d %5d %7.1f %10.3f %9.1f" % (name.split("(")[0][:44], bx, y, z, n / steps, tot / 1e6 / steps,
                                                   tot / n / 1e3))


if __name__ == "__main__":
    main()
