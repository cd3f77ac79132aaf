#!/usr/bin/env python3
"""Turn rocprofv3 result databases (rocpd sqlite) into the text summaries kept under
profiles/.  usage: rocprof_summary.py stats <db> | pmc <db>"""
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    print("# rocprofv3 --kernel-trace --stats  (%s)" % db)
    print("%-28s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.split("(")[0]
        print("%-28s %8d %14.1f %12.2f %6.2f%%" % (short, calls, tot, avg, pct))


def pmc(db):
    c = sqlite3.connect(db)
    print("# rocprofv3 --pmc  (%s)   value = per-dispatch average of the counter as reported" % db)
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
         "group by kernel_name, counter_name order by avg(value) desc")
    print("%-28s %-12s %8s %16s %12s" % ("kernel", "counter", "calls", "avg_value", "avg_dur_ns"))
    for name, cn, n, v, d in c.execute(q):
        print("%-28s %-12s %8d %16.2f %12.0f" % (name.split("(")[0], cn, n, v, d))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
