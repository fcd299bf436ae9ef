#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats table
(what `--stats` prints in CSV mode): calls, total/avg/min/max duration, share of GPU time.
usage: python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = db.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    out = [("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
    for name, n, tot, mn, mx in rows:
        short = name.split("(")[0][:90]
        out.append((short, n, round(tot / 1e6, 3), round(tot / n / 1e3, 2), round(mn / 1e3, 2), round(mx / 1e3, 2),
                    round(100.0 * tot / total, 2)))
    w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)
    print("# total kernel time %.3f ms over a span of %.3f ms, %d dispatches"
          % (total / 1e6, (span[1] - span[0]) / 1e6, sum(r[1] for r in rows)), file=sys.stderr)


if __name__ == "__main__":
    main()
