#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 --kernel-trace database in launch order: start (us from the step's first
dispatch), duration, gap to the previous dispatch's end on the same queue, queue, grid, kernel.  The step is the span
between two consecutive dispatches of MARKER (e.g. clip_adam).
usage: python tools/rocpd_sequence.py results.db MARKER_KERNEL_SUBSTR [which_step_from_end=2]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2]
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
    gx = "d.grid_size_x" if "grid_size_x" in cols else "d.grid_x"
    wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else "d.workgroup_x"
    rows = db.execute("select d.start, d.end, d.queue_id, %s/%s, s.kernel_name from rocpd_kernel_dispatch d join "
                      "rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % (gx, wx)).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[4]]
    a, b = marks[-back - 1], marks[-back]
    t0 = rows[a + 1][0]
    last_end = {}
    for st, en, q, blocks, name in rows[a + 1:b + 1]:
        gap = (st - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = en
        print("%9.1f %8.1f %7.1f  q%-3d %6d  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, q, blocks, name.split("(")[0][:90]))


if __name__ == "__main__":
    main()
