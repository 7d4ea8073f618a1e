#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (markdown), because `--stats` post-processing
is not always emitted.  Usage: rocpd_stats.py results.db [pmc]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                       "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % time |\n|---|---:|---:|---:|---:|---:|---:|")
    for r in rows[:25]:
        print("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (r[0][:90], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    if len(sys.argv) > 2:
        pe = [t for t in tabs if t.startswith("rocpd_pmc_event")]
        ip = [t for t in tabs if t.startswith("rocpd_info_pmc")]
        if pe and ip:
            q = ("select s.kernel_name, p.name, count(*), avg(e.value) from %s e join %s p on e.pmc_id=p.id join %s d on e.event_id=d.event_id "
                 "join %s s on d.kernel_id=s.id group by s.kernel_name, p.name order by 4 desc" % (pe[0], ip[0], kd, ks))
            print("\n| kernel | counter | dispatches | avg value |\n|---|---|---:|---:|")
            for r in cur.execute(q).fetchall()[:30]:
                print("| `%s` | %s | %d | %.1f |" % (r[0][:90], r[1], r[2], r[3]))


if __name__ == "__main__":
    main()
