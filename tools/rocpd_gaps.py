#!/usr/bin/env python3
"""GPU idle time inside the steady-state steps of a rocprofv3 --kernel-trace database: union of the busy intervals of
all queues, idle gaps between them, and which kernels the longest / most frequent gaps FOLLOW (the kernel after which
the GPU ran dry = where the host could not keep up).
usage: python tools/rocpd_gaps.py results.db MARKER_KERNEL_SUBSTR steps_to_use"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2]
    use = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rows = db.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    # one marker kernel per step (e.g. the optimizer): the last `use` steps
    lo, hi = marks[-use - 1], marks[-1]
    sel = rows[lo + 1:hi + 1]
    span = sel[-1][1] - sel[0][0]
    busy_end = sel[0][0]
    busy = 0
    gaps = []
    prev_name = None
    for st, en, name in sel:
        if st > busy_end:
            gaps.append((st - busy_end, prev_name, name))
            busy_end = st
        if en > busy_end:
            busy += en - busy_end
            busy_end = en
            prev_name = name
    idle = span - busy
    print("steps %d: span %.3f ms/step, busy %.3f ms/step, idle %.3f ms/step (%d gaps/step), kernel-time sum %.3f ms/step"
          % (use, span / use / 1e6, busy / use / 1e6, idle / use / 1e6, len(gaps) // use,
             sum(e - s for s, e, _ in sel) / use / 1e6))
    hist = defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        k = (a.split("(")[0][:40], b.split("(")[0][:40])
        hist[k][0] += 1
        hist[k][1] += g
    print("%-42s %-42s %8s %10s %8s" % ("after kernel", "before kernel", "gaps/st", "idle us/st", "avg us"))
    for (a, b), (n, t) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-42s %-42s %8.1f %10.1f %8.1f" % (a, b, n / use, t / use / 1e3, t / n / 1e3))
    # gap size histogram
    edges = [0, 2e3, 5e3, 10e3, 20e3, 50e3, 100e3, 1e9]
    for lo_, hi_ in zip(edges[:-1], edges[1:]):
        gs = [g for g, _, _ in gaps if lo_ <= g < hi_]
        print("gaps %6.0f-%6.0f us: %6.1f /step, %8.1f us/step" % (lo_ / 1e3, min(hi_, 1e6) / 1e3, len(gs) / use, sum(gs) / use / 1e3))


if __name__ == "__main__":
    main()
