#!/usr/bin/env python3
"""Aggregate PMC counters of a rocprofv3 rocpd database per kernel (sum and per-dispatch mean).
usage: python tools/rocpd_pmc.py <db> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [c[1] for c in db.execute("pragma table_info('rocpd_pmc_event')")]
    pmc_cols = [c[1] for c in db.execute("pragma table_info('rocpd_info_pmc')")]
    # event -> dispatch join: rocpd_pmc_event(event_id, pmc_id, value); rocpd_kernel_dispatch(event_id, kernel_id)
    q = ("select s.kernel_name, p.name, count(*), sum(e.value) from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on d.event_id = e.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1,2 order by 1,2")
    try:
        rows = db.execute(q).fetchall()
    except sqlite3.OperationalError as ex:
        print("schema:", cols, pmc_cols, ex)
        raise
    w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
    w.writerow(("kernel", "counter", "dispatches", "sum", "mean_per_dispatch"))
    for name, ctr, n, tot in rows:
        w.writerow((name.split("(")[0][:80], ctr, n, tot, tot / n))


if __name__ == "__main__":
    main()
